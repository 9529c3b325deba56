"""ctypes wrapper over oracle/_ref/libref_harness.so (the REAL reference, build container only).

TEST INFRASTRUCTURE: used by tools/make_golden.py and by the `-m "not gpu"` tests that are skipped when
oracle/_ref is absent.  Never imported by the product package, bench.py's timed path or the GPU tests.
"""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "_ref", "libref_harness.so")


def available():
    return os.path.exists(LIB)


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(LIB)
        _lib.ref_load.restype = C.c_void_p
        _lib.ref_load.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int]
        _lib.ref_fresnel.restype = C.c_float
        _lib.ref_fresnel.argtypes = [C.c_void_p, C.c_void_p, C.c_float]
        _lib.ref_refract.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
        _lib.ref_powf.restype = C.c_float
        _lib.ref_powf.argtypes = [C.c_float, C.c_float]
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class RefScene:
    """One reference Scene.  The reference keeps process-global flags, so use one live scene at a time."""

    def __init__(self, scene_path, width=-1, height=-1, cwd=ROOT, workers=None):
        self.h = C.c_void_p(lib().ref_load(cwd.encode(), scene_path.encode(), width, height))
        if not self.h:
            raise RuntimeError("ref_load failed")
        w, h, no, nl = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        lib().ref_dims(self.h, C.byref(w), C.byref(h), C.byref(no), C.byref(nl))
        self.width, self.height, self.n_objects, self.n_lights = w.value, h.value, no.value, nl.value
        if workers:
            lib().ref_set_workers(self.h, workers)

    def camera(self):
        scale, aspect = C.c_float(), C.c_float()
        m = np.zeros(16, np.float32)
        pos = np.zeros(3, np.float32)
        lib().ref_camera(self.h, C.byref(scale), C.byref(aspect), _p(m), _p(pos))
        return np.float32(scale.value), np.float32(aspect.value), m, pos

    def pass1(self):
        fb = np.zeros((self.height, self.width, 3), np.float32)
        lib().ref_pass1(self.h, _p(fb))
        return fb

    def ssaa(self, fb):
        fb = np.ascontiguousarray(fb.copy())
        lib().ref_ssaa(self.h, _p(fb))
        return fb

    def stats(self, fn):
        """Run fn() with the reference's statistics on; returns (result, [rays, boxTests, triTests])."""
        lib().ref_set_flag(b"collectStatistics", 1)
        lib().ref_stats_reset()
        r = fn()
        out = np.zeros(3, np.int64)
        lib().ref_stats(_p(out))
        lib().ref_set_flag(b"collectStatistics", 0)
        return r, out

    def probe(self, rays):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        n = rays.shape[0]
        out = np.zeros((n, 8), np.float32)
        col = np.zeros((n, 3), np.float32)
        lib().ref_probe(self.h, n, _p(rays), _p(out), _p(col))
        return out, col

    def skybox(self, d):
        d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
        out = np.zeros_like(d)
        for i in range(d.shape[0]):
            lib().ref_skybox(self.h, _p(d[i]), _p(out[i]))
        return out

    def illuminate(self, light, pts):
        pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 3)
        out = np.zeros((pts.shape[0], 8), np.float32)
        for i in range(pts.shape[0]):
            lib().ref_illuminate(self.h, light, _p(pts[i]), _p(out[i]))
        return out

    def bvh(self, obj_idx):
        cnt = np.zeros(5, np.int64)
        if lib().ref_bvh_counts(self.h, obj_idx, _p(cnt)) != 0:
            return None
        nn, nl, nr, md, nt = [int(x) for x in cnt]
        d = dict(bounds=np.zeros((nn, 6), np.float32), skip=np.zeros(nn, np.int32),
                 leaf_begin=np.zeros(nn, np.int32), leaf_count=np.zeros(nn, np.int32),
                 refs=np.zeros(nr, np.uint32))
        lib().ref_bvh_dump(self.h, obj_idx, _p(d["bounds"]), _p(d["skip"]), _p(d["leaf_begin"]),
                           _p(d["leaf_count"]), _p(d["refs"]))
        tris = np.zeros((nt, 30), np.float32)
        lib().ref_tris(self.h, obj_idx, _p(tris))
        d.update(tris=tris, n_nodes=nn, n_leaves=nl, n_refs=nr, max_depth=md, n_tris=nt)
        return d

    def save(self, fb, name_no_ext):
        fb = np.ascontiguousarray(fb, np.float32)
        return lib().ref_save(self.h, _p(fb), name_no_ext.encode())


def reflect(d, n):
    d = np.ascontiguousarray(d, np.float32); n = np.ascontiguousarray(n, np.float32)
    out = np.zeros(3, np.float32)
    lib().ref_reflect(_p(d), _p(n), _p(out))
    return out


def refract(d, n, ior):
    d = np.ascontiguousarray(d, np.float32); n = np.ascontiguousarray(n, np.float32)
    out = np.zeros(3, np.float32)
    lib().ref_refract(_p(d), _p(n), C.c_float(ior), _p(out))
    return out


def fresnel(d, n, ior):
    d = np.ascontiguousarray(d, np.float32); n = np.ascontiguousarray(n, np.float32)
    return np.float32(lib().ref_fresnel(_p(d), _p(n), C.c_float(ior)))


def normalize(v):
    v = np.ascontiguousarray(v, np.float32)
    out = np.zeros(3, np.float32)
    lib().ref_normalize(_p(v), _p(out))
    return out


def powf(x, y):
    return np.float32(lib().ref_powf(C.c_float(x), C.c_float(y)))
