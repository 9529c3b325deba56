"""Host-side logic without a GPU: the C-ABI library loads and exports every symbol of include/rtx.h, the
C++ loaders + BVH builder reproduce the oracle / the reference goldens, the flattened description is
consistent.  No compute calls are made here."""
import hashlib
import json
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def sha(a):
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_cabi_exports_every_declared_symbol(ra):
    hdr = open(os.path.join(ROOT, "include", "rtx.h")).read()
    declared = set(re.findall(r"\b(rtx_[a-z0-9_]+)\s*\(", hdr))
    listed, missing = ra.exported_symbols()
    assert not missing
    assert declared == set(listed), declared ^ set(listed)


def test_no_gpu_fails_loudly(ra):
    """Without a device the product path errors out (no CPU fallback).  Skipped on a GPU box."""
    try:
        n = ra.device_count()
    except ra.RtxError:
        return
    if n > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(ra.RtxError):
        ra.math_probe(0, np.ones(4, np.float32), 2.0)


@pytest.mark.parametrize("name", ["cfg1_simple_shapes", "cfg2_smooth_4k", "cfg2_smooth_25k", "cfg3_reflective_refractive",
                                  "cfg4_textured_256", "mixed_materials", "area_light"])
def test_host_loader_and_bvh_match_reference_golden(ra, name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    w, h = int(g["width"]), int(g["height"])
    s = ra.Scene("scenes/%s.scene" % name, w, h)
    scale, aspect, m, pos = s.camera()
    b = lambda a: np.ascontiguousarray(a, np.float32).view(np.uint32)
    assert b(scale) == b(g["cam_scale"]) and b(aspect) == b(g["cam_aspect"])
    assert np.array_equal(b(m), b(g["cam_matrix"])) and np.array_equal(b(pos), b(g["cam_pos"]))
    for i in range(s.n_objects):
        d = s.bvh(i)
        if d is None:
            assert "bvh%d_counts" % i not in g
            continue
        assert list(g["bvh%d_counts" % i]) == [d["n_nodes"], d["n_leaves"], d["n_refs"], d["max_depth"], d["n_tris"]]
        for k in ("bounds", "skip", "leaf_begin", "leaf_count", "refs", "tris"):
            assert sha(d[k]) == str(g["bvh%d_%s_sha1" % (i, k)]), k
        # structural invariants of the flat pre-order layout the kernels walk
        n = d["n_nodes"]
        leaf = d["leaf_count"] >= 0
        assert np.all(d["skip"][leaf] == np.arange(n)[leaf] + 1)
        assert np.all(d["skip"][~leaf] > np.arange(n)[~leaf] + 1) and d["skip"].max() == n
        lb = d["leaf_begin"][leaf]
        assert np.array_equal(lb, np.concatenate([[0], np.cumsum(d["leaf_count"][leaf])[:-1]]))


def test_host_250k_bvh_digest(ra):
    from rendering_amd import assets
    d = json.load(open(os.path.join(GOLD, "cfg2_smooth_250k_digest.json")))
    assets.ensure(["bumpy_250k.obj"])
    s = ra.Scene("scenes/cfg2_smooth_250k.scene", 64, 64)
    b = s.bvh(1)
    assert [b["n_nodes"], b["n_leaves"], b["n_refs"], b["max_depth"], b["n_tris"]] == d["counts"]
    for k, v in d["sha1"].items():
        assert sha(b[k]) == v, k


def test_host_save_bmp_matches_reference_bytes(ra, tmp_path):
    g = np.load(os.path.join(GOLD, "units.npz"))
    s = ra.Scene("scenes/cfg1_simple_shapes.scene", 8, 4)
    assert s.save_bmp(g["quant_fb"], str(tmp_path / "q")) == 0
    assert open(str(tmp_path / "q.bmp"), "rb").read() == g["quant_bmp"].tobytes()
