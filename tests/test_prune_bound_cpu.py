"""CPU: the error bound the prune records rest on (DESIGN_HISTORY.md 3.1c): a hit the reference accepts lies within
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


def test_source_records_bound_accepted_hits():
    """DESIGN_HISTORY.md 3.1d: with the source certificate (rays that start at the camera / end at a point light) an accepted hit lies within
    216 dmax ainf P_S + 2^-17 (ainf + |orig|) of its triangle's box, P_S from the function the device runs (rtx_source_p_probe); pairs
    without a certificate fall back to Pgen and stay within the unconditional bound."""
    from tools.research.src_bound_check import run
    rng = np.random.default_rng(11)
    certified = 0
    for cam in (True, False):
        for mode in ("graze", "sliver", "generic"):
            ratio, ncert, off_cert, ratio_cert = run(40000, mode, cam, rng, quiet=True)
            assert ratio < 0.25, "%s %s: an accepted hit at %.3f of the bound" % ("camera" if cam else "light", mode, ratio)
            assert ratio_cert < 0.25
            certified += ncert
    assert certified > 5000, "the sample no longer exercises the certificate"
    # culling off (objects.cpp:75-79: |det_c| < 1e-8 rejects; back faces are accepted): the same bound, the source on either side of the plane --
    # what lets the culling-off walk use the source copies of the prune records as well (pruneEval8<.., false>)
    back = 0
    for cam in (True, False):
        for mode in ("graze", "sliver", "generic"):
            ratio, ncert, off_cert, ratio_cert = run(40000, mode, cam, rng, quiet=True, cull=False)
            assert ratio < 0.25 and ratio_cert < 0.25, "culling off, %s %s: an accepted hit at %.3f of the bound" % ("camera" if cam else "light", mode, max(ratio, ratio_cert))
            back += ncert
    assert back > 5000


def test_source_p_never_exceeds_pgen_and_needs_height():
    import rendering_amd as RA
    rng = np.random.default_rng(3)
    n = 2000
    v0 = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    e1 = (rng.normal(size=(n, 3)) * 0.02).astype(np.float32); e2 = (rng.normal(size=(n, 3)) * 0.02).astype(np.float32)
    pgen = np.abs(e1.astype(np.float64)).sum(1) * np.abs(e2.astype(np.float64)).sum(1)
    for cam in (True, False):
        p = RA.source_p_probe(v0, e1, e2, [0.3, 4.0, -2.0], 1e-4, cam).astype(np.float64)
        assert (p <= pgen * (1 + 1e-5) + 1e-36).all() and (p > 0).all()
        assert (p < 0.05 * pgen).mean() > 0.9           # a source well off the planes: the certificate holds nearly everywhere
    # a source IN the plane of a triangle has no certificate; so have big triangles and degenerate ones
    S = (v0[0].astype(np.float64) + 0.3 * e1[0] + 5.0 * e2[0])
    assert RA.source_p_probe(v0[:1], e1[:1], e2[:1], S, 0.0, True)[0] >= np.float32(pgen[0])
    big = RA.source_p_probe(v0[:1], e1[:1] * 50, e2[:1] * 50, [0.3, 4.0, -2.0], 0.0, True)[0]
    assert big >= np.float32(pgen[0] * 2500 * 0.999)
    z = RA.source_p_probe(v0[:1], e1[:1] * 0, e2[:1], [0.3, 4.0, -2.0], 0.0, True)[0]
    assert z == z and z >= 0


def test_prune_bounds_on_the_headline_frame_real_data():
    """over rows of the headline frame the hits the reference accepts (CPU oracle: closest hits of primary rays and of
    the shadow rays towards the three lights) stay within the inflation the prune records allow -- with Pgen and with the sources' P_S."""
    from tools.research.rho_real_data import main
    import io
    res = main(stride=1024, S=4096, out=io.StringIO())
    assert len(res) == 4
    for ratio_src, ratio_gen, n in res:
        assert n > 500 and ratio_src < 0.25 and ratio_gen < 0.25
