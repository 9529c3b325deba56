"""GPU: the HIP path against the COMMITTED golden vectors (tests/golden/*.npz, produced from the real reference by
tools/make_golden.py) -- no oracle in between.  Bit-exact float framebuffers (pass 1, post-SSAA), per-ray records,
64-bit statistics, camera constants and acceleration-structure digests (device-built)."""
import hashlib
import os

import numpy as np
import pytest

from tests.util_rays import probe_rays

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SCENES = ["cfg1_simple_shapes", "cfg2_smooth_4k", "cfg2_smooth_25k", "cfg3_reflective_refractive", "cfg4_textured_256",
          "mixed_materials", "area_light", "coincident"]


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def sha(a):
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize("name", SCENES)
def test_hip_path_matches_reference_golden(ra, name):
    from rendering_amd import assets
    g = np.load(os.path.join(GOLD, name + ".npz"))
    for item in str(g["assets_md5"]).split(";"):
        if item:
            n, md5 = item.split("=")
            assert assets.md5(n) == md5, "generated asset %s differs from the one the golden was made with" % n
    w, h = int(g["width"]), int(g["height"])
    ra.set_ac_build("device")          # (the default picks the host builder for meshes this small)
    try:
        s = ra.Scene("scenes/%s.scene" % name, w, h)
    finally:
        ra.set_ac_build("auto")
    scale, aspect, m, pos = s.camera()
    assert bits(scale) == bits(g["cam_scale"]) and bits(aspect) == bits(g["cam_aspect"])
    assert np.array_equal(bits(m), bits(g["cam_matrix"])) and np.array_equal(bits(pos), bits(g["cam_pos"]))
    s.counters_enable(True)
    s.counters_reset()
    fb1 = s.render_host(ssaa=False)
    st = s.counters()
    s.counters_enable(False)
    assert np.array_equal(bits(fb1), bits(g["pass1"]))
    assert np.array_equal(st, g["pass1_stats"])          # rays, box tests, triangle tests (reference semantics)
    fb2 = s.render_host(ssaa=True)
    d = (bits(fb2) != bits(g["ssaa"])).any(-1)
    d[0, :] = False; d[:, 0] = False                     # uninitialised Sobel border in the reference
    assert not d.any()
    hits, col = s.cast_rays(probe_rays(1024))
    assert np.array_equal(bits(hits), bits(g["probe_hits"]))
    assert np.array_equal(bits(col), bits(g["probe_colours"]))
    for i in range(s.n_objects):
        b = s.bvh(i)
        if b is None:
            assert "bvh%d_counts" % i not in g
            continue
        assert b["built_on_device"]
        assert list(g["bvh%d_counts" % i]) == [b["n_nodes"], b["n_leaves"], b["n_refs"], b["max_depth"], b["n_tris"]]
        for k in ("bounds", "skip", "leaf_begin", "leaf_count", "refs", "tris"):
            assert sha(b[k]) == str(g["bvh%d_%s_sha1" % (i, k)]), k


def test_headline_250k_mesh_matches_reference_digest(ra):
    """The headline mesh pinned to the REFERENCE directly on the GPU (no oracle in between): device-built acceleration
    structure sha1s, the 128x128 pass-1 framebuffer sha1 and the 64-bit statistics recorded from oracle/_ref
    (tests/golden/cfg2_smooth_250k_digest.json, tools/make_golden.py)."""
    import json
    from rendering_amd import assets
    assets.ensure(["bumpy_250k.obj"])
    d = json.load(open(os.path.join(GOLD, "cfg2_smooth_250k_digest.json")))
    assert assets.md5("bumpy_250k.obj") == d["asset_md5"]
    s = ra.Scene("scenes/cfg2_smooth_250k.scene", 128, 128)
    b = s.bvh(1)
    assert b["built_on_device"]
    assert d["counts"] == [b["n_nodes"], b["n_leaves"], b["n_refs"], b["max_depth"], b["n_tris"]]
    for k, v in d["sha1"].items():
        assert sha(b[k]) == v, k
    s.counters_enable(True)
    s.counters_reset()
    fb = s.render_host(ssaa=False)
    st = s.counters()
    s.counters_enable(False)
    assert [int(x) for x in st] == d["pass1_128x128_stats"]
    assert sha(fb) == d["pass1_128x128_sha1"]


def test_device_vector_helpers_match_reference_units(ra):
    """reflect / refract / fresnel / normalize evaluated ON THE DEVICE against the vectors recorded from the reference
    (tests/golden/units.npz: scene.cpp:672-722, geometry.h:104-112) -- the pieces of the shading path that the frame tests
    only see through colours."""
    g = np.load(os.path.join(GOLD, "units.npz"))
    d, n = g["d"], g["n"]
    assert np.array_equal(bits(ra.vec_probe(0, d, n)), bits(g["reflect"]))
    for ior in ("1.4", "1", "0.7", "2.5"):
        assert np.array_equal(bits(ra.vec_probe(1, d, n, float(ior))), bits(g["refract_" + ior])), "refract " + ior
        assert np.array_equal(bits(ra.vec_probe(2, d, n, float(ior))[:, 0]), bits(g["fresnel_" + ior])), "fresnel " + ior
    assert np.array_equal(bits(ra.vec_probe(3, g["normalize_in"])), bits(g["normalize_out"]))
