"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU oracle on the same inputs.

Bar (BASELINE.json north_star): bit-exact float framebuffers / BMP bytes on integer-coordinate scenes
(spheres/planes, untextured meshes); <= 1e-4 per-channel RMSE on the textured scene, where the reference
itself is only reproducible to ~1 ULP (normal-map in-place normalisation race).
"""
import hashlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BITEXACT = [
    ("cfg1_simple_shapes", 512, 512),
    ("cfg3_reflective_refractive", 480, 272),
    ("cfg2_smooth_4k", 320, 240),
    ("cfg2_smooth_25k", 200, 160),
    ("mixed_materials", 320, 240),
    ("cfg4_textured_256", 256, 256),
    ("area_light", 320, 240),
    ("coincident", 320, 240),     # every hit is an exact t tie between two triangles with different normals
    ("uv_out_of_range", 256, 256),   # texture coordinates < 0 and > 1: both sides clamp to the edge texel
]


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def ndiff(a, b):
    return int((bits(a) != bits(b)).any(-1).sum())


@pytest.mark.parametrize("name,w,h", BITEXACT)
def test_pass1_and_ssaa_bit_exact(ra, oracle, name, w, h):
    path = "scenes/%s.scene" % name
    o = oracle.OracleScene(path, w, h)
    g = ra.Scene(path, w, h)
    ref1 = o.pass1()
    got1 = g.render_host(ssaa=False)
    assert ndiff(ref1, got1) == 0, "pass-1 framebuffer differs in %d pixels" % ndiff(ref1, got1)
    # last row / column never rendered (scene.cpp:369-372)
    assert not got1[-1].any() and not got1[:, -1].any()
    ref2 = o.ssaa(ref1)
    got2 = g.render_host(ssaa=True)
    assert ndiff(ref2, got2) == 0, "post-SSAA framebuffer differs in %d pixels" % ndiff(ref2, got2)
    assert oracle.encode_bmp(ref2) == oracle.encode_bmp(got2)


def test_cfg1_bmp_md5_matches_reference_cli(ra, oracle):
    """cfg1 at 512x512: md5 of the BMP the real reference CLI writes (reproduced in this repo's
    build container with oracle/_ref/render_ref)."""
    g = ra.Scene("scenes/cfg1_simple_shapes.scene", 512, 512)
    fb = g.render_host(ssaa=True)
    assert hashlib.md5(oracle.encode_bmp(fb)).hexdigest() == "0e5f3b78e230e36d9bc9ed8fcdfa6fd3"


@pytest.mark.parametrize("name", ["cfg1_simple_shapes", "cfg2_smooth_4k", "mixed_materials", "cfg4_textured_256",
                                  "cfg3_reflective_refractive", "area_light"])
def test_probe_rays_bit_exact(ra, oracle, name):
    """4k pseudo-random + grazing rays: hit record {hit, object, triangle, t, u, v} and castRay colour."""
    from tests.util_rays import probe_rays
    path = "scenes/%s.scene" % name
    o = oracle.OracleScene(path, 64, 64)
    g = ra.Scene(path, 64, 64)
    rays = probe_rays(4096)
    rh, rc = o.probe(rays)
    gh, gc = g.cast_rays(rays)
    assert np.array_equal(bits(rh), bits(gh)), "hit records differ at %s" % np.argwhere(bits(rh) != bits(gh))[:5]
    assert np.array_equal(bits(rc), bits(gc))


def test_device_math_matches_host(ra, oracle):
    """powf (restated glibc), IEEE divide, sqrtf and the fp64 normalize factor on the device vs the host."""
    rng = np.random.default_rng(7)
    x = np.concatenate([rng.random(200000, dtype=np.float32), np.float32(1) - rng.random(50000, dtype=np.float32) * np.float32(1e-3),
                        rng.random(50000, dtype=np.float32) * np.float32(1e-6), np.array([0, 1, 1e-30, 1e-40, 2, 10, np.inf], np.float32)])
    for y in (5.0, 10.0, 2.0, 0.5, 20.0, 64.0, 3.7):
        got = ra.math_probe(0, x, np.float32(y))
        ref = np.array([oracle.powf(float(a), y) for a in x[:20000]], np.float32)
        assert np.array_equal(bits(got[:20000]), bits(ref)), "powf(x,%g)" % y
    xs = (rng.random(300000, dtype=np.float32) - np.float32(0.5)) * np.float32(8)
    xs[:5] = [0.0, -0.0, 1e-42, 3e38, -1e-39]
    with np.errstate(all="ignore"):
        assert np.array_equal(bits(ra.math_probe(1, xs)), bits(np.float32(1) / xs))
        pos = np.abs(xs)
        assert np.array_equal(bits(ra.math_probe(2, pos)), bits(np.sqrt(pos)))
        inv = (1.0 / np.sqrt(pos.astype(np.float64))).astype(np.float32)
        assert np.array_equal(bits(ra.math_probe(3, pos)), bits(inv))
        ys = (rng.random(300000, dtype=np.float32) - np.float32(0.5)) * np.float32(3)
        assert np.array_equal(bits(ra.math_probe(4, xs, ys)), bits(xs / ys))


def test_normalize_factor_every_mantissa(ra):
    """Vec3::normalize's factor (float)(1.0 / sqrt((double)len2)) (geometry.h:104-112) on the device -- the exact fp32 fast path of invLenD with its fp64
    fallback -- against the expression itself for EVERY float mantissa, both exponent parities, at exponents inside, at the ends of and outside the fast
    path's range, plus zeros, denormals, inf and NaN (tools/research/rsqrt_exhaustive.c is the same enumeration on the CPU for every starting value v_rsq_f32
    could return)."""
    m = np.arange(1 << 23, dtype=np.uint32)
    for e in (127, 128, 126, 127 - 60, 127 - 61, 128 + 59, 127 + 61, 1, 254, 127 + 20, 127 - 33):
        x = (np.uint32(e << 23) | m).view(np.float32)
        want = (1.0 / np.sqrt(x.astype(np.float64))).astype(np.float32)
        got = ra.math_probe(3, x)
        bad = np.nonzero(bits(got) != bits(want))[0]
        assert bad.size == 0, "exponent %d: %d of 2^23 differ, first x = %r: %r != %r" % (e, bad.size, x[bad[0]], got[bad[0]], want[bad[0]])
    sp = np.array([0.0, 1e-45, 1e-39, 1.17549435e-38, 3.4028235e38, np.inf, np.nan, 1.0, 4.0, 0.25], np.float32)
    with np.errstate(all="ignore"):
        want = (1.0 / np.sqrt(sp.astype(np.float64))).astype(np.float32)
    got = ra.math_probe(3, sp)
    ok = (bits(got) == bits(want)) | (np.isnan(got) & np.isnan(want))
    assert ok.all(), (sp[~ok], got[~ok], want[~ok])


def test_counters_match_reference_semantics(ra, oracle):
    """64-bit rays / box tests / triangle tests under reference traversal semantics (stats.h)."""
    path = "scenes/cfg2_smooth_4k.scene"
    o = oracle.OracleScene(path, 160, 120)
    g = ra.Scene(path, 160, 120)
    _, ref = o.stats(lambda: o.pass1())
    g.counters_enable(True)
    g.counters_reset()
    g.render_host(ssaa=False)
    got = g.counters()
    g.counters_enable(False)
    assert np.array_equal(ref, got), (ref, got)


@pytest.mark.parametrize("local_below", ["0", "4000000000"])
def test_ssaa_list_modes_bit_exact(ra, oracle, monkeypatch, local_below):
    """Both layouts of the SSAA flagged-pixel list (dense = full waves, tile-local = padded) give the reference's
    frame; the knob forces the mode the device would otherwise pick from the number of flagged pixels."""
    monkeypatch.setenv("RTX_SSAA_LOCAL_BELOW", local_below)
    for name, w, h in (("cfg2_smooth_4k", 320, 240), ("cfg1_simple_shapes", 256, 256)):
        path = "scenes/%s.scene" % name
        o = oracle.OracleScene(path, w, h)
        g = ra.Scene(path, w, h)
        ref = o.ssaa(o.pass1())
        got = g.render_host(ssaa=True)
        assert ndiff(ref, got) == 0


@pytest.mark.parametrize("spread_slots", ["1048576", "4096", "0"])
def test_ssaa_four_pixel_waves_and_slot_budget(ra, oracle, monkeypatch, spread_slots):
    """Tile-local SSAA list with EVERY tile classified as very slow (knob: threshold of 1 tick), so that every tile asks
    for the 4-pixels-per-wave layout (4x the slots): with a full, a partly and a fully exhausted slot budget the list
    never overruns (tiles over the budget are packed normally) and the frame is the reference's."""
    monkeypatch.setenv("RTX_SSAA_LOCAL_BELOW", "4000000000")
    monkeypatch.setenv("RTX_SSAA_HEAVY_TICKS", "1")
    monkeypatch.setenv("RTX_SSAA_SPREAD_SLOTS", spread_slots)
    for name, w, h in (("cfg2_smooth_4k", 320, 240), ("cfg3_reflective_refractive", 240, 136)):
        path = "scenes/%s.scene" % name
        o = oracle.OracleScene(path, w, h)
        g = ra.Scene(path, w, h)
        ref = o.ssaa(o.pass1())
        got = g.render_host(ssaa=True)
        assert ndiff(ref, got) == 0


@pytest.mark.parametrize("sparse_below", ["0", "4000000000"])
@pytest.mark.parametrize("heavy_ticks", ["1", "25000"])
def test_ssaa_sparse_layout_bit_exact(ra, oracle, monkeypatch, sparse_below, heavy_ticks):
    """The "sparse" layout of the tile-local SSAA list (rtxSsaaCountKernel: 4 pixels per wave for the tiles that were slow in pass 1, ONE
    for the very slow ones, chosen when a launch has fewer 16-pixel items than half its waves) forced on and off, with every tile classified
    as very slow (threshold of 1 tick: one-pixel waves everywhere) and with the product's threshold: the frame is the reference's."""
    monkeypatch.setenv("RTX_SSAA_LOCAL_BELOW", "4000000000")
    monkeypatch.setenv("RTX_SSAA_SPARSE_BELOW", sparse_below)
    monkeypatch.setenv("RTX_SSAA_HEAVY_TICKS", heavy_ticks)
    for name, w, h in (("cfg2_smooth_4k", 320, 240), ("cfg3_reflective_refractive", 240, 136)):
        path = "scenes/%s.scene" % name
        o = oracle.OracleScene(path, w, h)
        g = ra.Scene(path, w, h)
        ref = o.ssaa(o.pass1())
        for _ in range(2):          # (the second frame knows the tile costs of the first)
            got = g.render_host(ssaa=True)
            assert ndiff(ref, got) == 0


def test_event_pool_does_not_grow(ra):
    """Launch timing keeps one event pair per kernel unless rtx_kernel_time_reset asked for accumulation."""
    g = ra.Scene("scenes/cfg1_simple_shapes.scene", 64, 64)
    for _ in range(50):
        g.render_host(ssaa=True)
    assert g.kernel_time_stats(0)[0] == 1 and g.kernel_time_stats(2)[0] == 1
    g.kernel_time_reset()
    for _ in range(3):
        g.render_host(ssaa=True)
    n, ms = g.kernel_time_stats(0)
    assert n == 3 and ms > 0


def test_tile_cost_map(ra):
    """rtx_tile_cost_read: one entry per 8x8 tile of the frame; tiles on the mesh cost more than background tiles."""
    g = ra.Scene("scenes/cfg2_smooth_4k.scene", 160, 120)
    g.render_host(ssaa=False)
    c = g.tile_cost()
    assert c.shape == (15, 20) and c.dtype == np.uint32
    assert c.min() > 0 and c[7, 10] > c[0, 0]
    # the SSAA launch records its slowest work item per tile: exactly the tiles with flagged pixels have one
    import torch
    fb = torch.zeros((120, 160, 3), dtype=torch.float32, device="cuda"); mask = torch.zeros((120, 160), dtype=torch.uint8, device="cuda")
    g.render_pass1(fb); g.sobel(fb, mask); g.render_ssaa(mask, fb)
    torch.cuda.synchronize()
    it = g.ssaa_item_cost()
    flagged = mask.cpu().numpy().reshape(15, 8, 20, 8).sum((1, 3)) > 0
    assert it.shape == (15, 20) and flagged.any() and np.array_equal(it > 0, flagged)
    assert np.array_equal(g.tile_cost(), c) or g.tile_cost().min() > 0      # (the first half is still the pass-1 map)


def test_surface_rays_250k_bit_exact(ra, oracle):
    """Stress of the leaf certificates (DESIGN_HISTORY.md 3.3) on the headline mesh: 60k rays that START ON the surface
    (like shadow / reflection rays: most of the mesh lies behind them) in pseudo-random directions, plus rays from
    inside the mesh, far outside, axis-aligned and grazing -- hit records and colours against the oracle."""
    from rendering_amd import assets
    assets.ensure(["bumpy_250k.obj"])
    path = "scenes/cfg2_smooth_250k.scene"
    o = oracle.OracleScene(path, 64, 64)
    g = ra.Scene(path, 64, 64)
    b = o.bvh(1)
    tris = b["tris"]
    rng = np.random.default_rng(0x5EED)
    n = 60000
    idx = rng.integers(0, tris.shape[0], n)
    w = rng.random((n, 3)).astype(np.float32)
    w /= w.sum(1, keepdims=True)
    p = (tris[idx, 0:3] * w[:, :1] + tris[idx, 3:6] * w[:, 1:2] + tris[idx, 6:9] * w[:, 2:3]).astype(np.float32)
    nrm = np.cross(tris[idx, 3:6] - tris[idx, 0:3], tris[idx, 6:9] - tris[idx, 0:3])
    nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-30)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    org = p + (nrm * np.float32(1e-4) * np.where(rng.random((n, 1)) < 0.7, 1, -1)).astype(np.float32)
    rays = np.concatenate([org, d], 1).astype(np.float32)
    rays[0:2000, 0:3] = np.float32([0, 0, -3]) + rng.normal(size=(2000, 3)).astype(np.float32) * np.float32(0.2)   # inside
    rays[2000:4000, 0:3] = rng.normal(size=(2000, 3)).astype(np.float32) * np.float32(50)                         # far away
    rays[4000:4600, 3:6] = np.float32([[1, 0, 0], [0, 1, 0], [0, 0, 1], [-1, 0, 0], [0, -1, 0], [0, 0, -1]] * 100)
    rays[4600:5200, 4] = np.float32(1e-9)
    rh, rc = o.probe(rays)
    gh, gc = g.cast_rays(rays)
    bad = np.argwhere((bits(rh) != bits(gh)).any(1))
    assert len(bad) == 0, "hit records differ for %d rays, first %s" % (len(bad), bad[:5].ravel())
    assert np.array_equal(bits(rc), bits(gc))
    assert (rh[:, 0] > 0).mean() > 0.3            # the set does exercise real hits
