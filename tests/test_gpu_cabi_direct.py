"""GPU: the C ABI driven directly (hand-made rtx_scene_desc, no host loader): a two-leaf mesh whose leaf boxes are NOT
nested in the root box -- what a foreign caller may pass.  The library must then keep to the walk that tests every
ancestor's box (the two-levels-at-a-time walk is exact for nested boxes only); hits are compared with a plain numpy
statement of the reference's semantics (objects.cpp:534-631, 59-95)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
f32 = np.float32


class View(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("bias", C.c_float), ("max_ray_depth", C.c_int32), ("background", C.c_float * 3),
                ("flags", C.c_uint32), ("cam_pos", C.c_float * 3), ("cam_matrix", C.c_float * 16), ("scale", C.c_float), ("aspect", C.c_float)]


class Obj(C.Structure):
    _fields_ = [("type", C.c_int32), ("material", C.c_int32), ("pos", C.c_float * 3), ("color", C.c_float * 3), ("ior", C.c_float), ("ambient", C.c_float),
                ("diffuse", C.c_float), ("specular", C.c_float), ("n_specular", C.c_float), ("radius2", C.c_float), ("normal", C.c_float * 3), ("mesh", C.c_int32)]


class Mesh(C.Structure):
    _fields_ = [("n_nodes", C.c_uint32), ("n_refs", C.c_uint32), ("n_tris", C.c_uint32), ("node_bounds", C.c_void_p), ("node_skip", C.c_void_p),
                ("leaf_begin", C.c_void_p), ("leaf_count", C.c_void_p), ("refs", C.c_void_p), ("tri_pos", C.c_void_p), ("tri_nrm", C.c_void_p),
                ("tri_uv", C.c_void_p), ("tri_tb", C.c_void_p), ("diffuse_w", C.c_uint32), ("diffuse_h", C.c_uint32), ("diffuse_map", C.c_void_p),
                ("normal_w", C.c_uint32), ("normal_h", C.c_uint32), ("normal_map", C.c_void_p), ("specular_w", C.c_uint32), ("specular_h", C.c_uint32),
                ("specular_map", C.c_void_p)]


class Desc(C.Structure):
    _fields_ = [("view", View), ("n_objects", C.c_uint32), ("objects", C.c_void_p), ("n_meshes", C.c_uint32), ("meshes", C.c_void_p),
                ("n_lights", C.c_uint32), ("lights", C.c_void_p), ("sky_w", C.c_uint32), ("sky_h", C.c_uint32), ("sky", C.c_void_p * 6)]


def slab(o, d, lo, hi):
    with np.errstate(all="ignore"):
        inv = f32(1) / d
        s = inv < 0
        bmin = np.where(s, hi, lo); bmax = np.where(s, lo, hi)
        tmn = (bmin - o) * inv; tmx = (bmax - o) * inv
        tmin, tmax = tmn[:, 0].copy(), tmx[:, 0].copy()
        fail = (tmin > tmx[:, 1]) | (tmn[:, 1] > tmax)
        tmin = np.where(tmn[:, 1] > tmin, tmn[:, 1], tmin); tmax = np.where(tmx[:, 1] < tmax, tmx[:, 1], tmax)
        fail |= (tmin > tmx[:, 2]) | (tmn[:, 2] > tmax)
    return ~fail


def mt(o, d, a, b, c):
    e1, e2 = b - a, c - a
    p = np.stack([d[:, 1] * e2[2] - d[:, 2] * e2[1], d[:, 2] * e2[0] - d[:, 0] * e2[2], d[:, 0] * e2[1] - d[:, 1] * e2[0]], 1).astype(f32)
    det = (e1[0] * p[:, 0] + e1[1] * p[:, 1] + e1[2] * p[:, 2]).astype(f32)
    ok = ~(det.astype(np.float64) < 1e-8)
    with np.errstate(all="ignore"):
        inv = f32(1) / det
        t = o - a
        u = ((t[:, 0] * p[:, 0] + t[:, 1] * p[:, 1] + t[:, 2] * p[:, 2]).astype(f32) * inv).astype(f32)
        ok &= ~((u < 0) | (u > 1))
        q = np.stack([t[:, 1] * e1[2] - t[:, 2] * e1[1], t[:, 2] * e1[0] - t[:, 0] * e1[2], t[:, 0] * e1[1] - t[:, 1] * e1[0]], 1).astype(f32)
        v = ((d[:, 0] * q[:, 0] + d[:, 1] * q[:, 1] + d[:, 2] * q[:, 2]).astype(f32) * inv).astype(f32)
        ok &= ~((v < 0) | ((u + v).astype(f32) > 1))
        tt = ((e2[0] * q[:, 0] + e2[1] * q[:, 1] + e2[2] * q[:, 2]).astype(f32) * inv).astype(f32)
        ok &= ~(tt < 0)
    return ok, tt


@pytest.mark.parametrize("nested", [False, True])
def test_hand_made_mesh_with_boxes_that_are_not_nested(ra, nested):
    rtx, _ = ra.load()
    # two triangles in the plane z = -3, facing +z; leaf 1 holds triangle 0, leaf 2 triangle 1
    tri = np.array([[-1, -1, -3, 1, -1, -3, 0, 1, -3], [0.5, -1, -3, 2.5, -1, -3, 1.5, 1, -3]], f32)
    if nested:
        bounds = np.array([[-1, -1, -3.1, 2.5, 1, -2.9], [-1, -1, -3.1, 1, 1, -2.9], [0.5, -1, -3.1, 2.5, 1, -2.9]], f32)
    else:
        # the root box covers x <= 1.2 only, the second leaf's box sticks out of it; the first leaf's box is cut short
        bounds = np.array([[-1, -1, -3.1, 1.2, 1, -2.9], [-1, -0.5, -3.1, 1, 1, -2.9], [0.5, -1, -3.2, 2.5, 1, -2.8]], f32)
    skip = np.array([3, 2, 3], np.int32); lb = np.array([-1, 0, 1], np.int32); lc = np.array([-1, 1, 1], np.int32)
    refs = np.array([0, 1], np.uint32)
    nrm = np.tile(np.array([0, 0, 1], f32), (2, 3)).astype(f32); uv = np.zeros((2, 6), f32)
    m = Mesh(3, 2, 2, bounds.ctypes.data, skip.ctypes.data, lb.ctypes.data, lc.ctypes.data, refs.ctypes.data, tri.ctypes.data, nrm.ctypes.data,
             uv.ctypes.data, None, 0, 0, None, 0, 0, None, 0, 0, None)
    ob = Obj(3, 0, (C.c_float * 3)(0, 0, 0), (C.c_float * 3)(1, 1, 1), 1.4, 0.1, 0.1, 1.0, 5.0, 0.0, (C.c_float * 3)(0, 0, 0), 0)
    v = View(64, 64, 1e-4, 2, (C.c_float * 3)(0, 0, 0), 1, (C.c_float * 3)(0, 0, 0), (C.c_float * 16)(1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1), 0.57735026, 1.0)
    d = Desc(v, 1, C.addressof(ob), 1, C.addressof(m), 0, None, 0, 0, (C.c_void_p * 6)())
    h = C.c_void_p()
    assert rtx.rtx_scene_create(C.byref(d), 0, C.byref(h)) == 0, rtx.rtx_last_error()
    try:
        rng = np.random.default_rng(5)
        n = 4096
        o = np.zeros((n, 3), f32); o[:, :2] = rng.uniform(-0.3, 0.3, (n, 2)).astype(f32)
        tgt = np.stack([rng.uniform(-1.3, 2.8, n), rng.uniform(-1.2, 1.2, n), np.full(n, -3.0)], 1).astype(f32)
        dd = (tgt - o).astype(f32)
        dd = (dd / np.linalg.norm(dd, axis=1)[:, None]).astype(f32)
        dd[::7, 0] = 0          # some exactly axis-parallel components (1/dir = inf: the NaN-capable form of the box test)
        rays = np.ascontiguousarray(np.concatenate([o, dd], 1), f32)
        hits = np.zeros((n, 8), f32); col = np.zeros((n, 3), f32)
        assert rtx.rtx_cast_rays(h, n, rays.ctypes.data, hits.ctypes.data, col.ctypes.data) == 0, rtx.rtx_last_error()
        # the reference: root box, then leaf 1, then leaf 2; strict <, first wins
        root = slab(o, dd, bounds[0, :3], bounds[0, 3:])
        best = np.full(n, np.finfo(f32).max, f32); btri = np.full(n, -1)
        for leaf, t_idx in ((1, 0), (2, 1)):
            reach = root & slab(o, dd, bounds[leaf, :3], bounds[leaf, 3:])
            ok, t = mt(o, dd, tri[t_idx, 0:3], tri[t_idx, 3:6], tri[t_idx, 6:9])
            upd = reach & ok & (t < best)
            best = np.where(upd, t, best); btri = np.where(upd, t_idx, btri)
        assert np.array_equal(hits[:, 0] != 0, btri >= 0)
        hit = btri >= 0
        assert np.array_equal(hits[hit, 2].astype(int), btri[hit])
        assert np.array_equal(hits[hit, 3].view(np.uint32), best[hit].view(np.uint32))
        assert hit.sum() > 500 and (~hit).sum() > 500
        if not nested:
            # rays that hit triangle 1 outside the root box must NOT be hits (the reference never gets there)
            outside = (o[:, 0] + dd[:, 0] * (-3 / dd[:, 2]) > 1.3) & (np.abs(dd[:, 0]) > 0)
            assert outside.sum() > 100 and not hit[outside].any()
    finally:
        rtx.rtx_scene_destroy(h)
