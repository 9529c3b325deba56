"""CPU: the error bound the prune records rest on: a hit the reference accepts lies within
36 u dmax ainf s1 s2 / det_c of its triangle's box -- and NOT within a small fixed margin: for rays nearly in the plane of a
triangle the reference's fp32 arithmetic (objects.cpp:59-95, restated in numpy, no FMA) accepts hits far off the triangle."""
import numpy as np


def test_accepted_hits_lie_within_the_bound_and_not_within_an_ulp_margin():
    from tools.research.rho_check import run
    rng = np.random.default_rng(7)
    worst_off = 0.0
    for mode in ("graze", "sliver", "generic"):
        ratio, accepted, off = run(300000, mode, rng, quiet=True)
        assert accepted > 1000, mode
        assert ratio < 0.25, "%s: an accepted hit at %.3f of the bound (the margin of 4 is gone)" % (mode, ratio)
        worst_off = max(worst_off, off)
    assert worst_off > 1e-3, "the adversarial pairs no longer reach hits that lie visibly off their triangle"
