"""ctypes wrapper over oracle/liboracle.so -- the travelling CPU oracle (TEST INFRASTRUCTURE).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; the product
package (rendering_amd/) never does.  Same method names as tools/ref_harness.RefScene so the two can be
compared generically.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "liboracle.so")


def build(force=False):
    """Compile the oracle (and, where /root/reference exists, oracle/_ref).  Building is not using."""
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(os.path.join(HERE, "rt_oracle.cpp")):
        subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])
    if os.path.exists("/root/reference/src/scene.cpp"):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB)
        _lib.orc_load.restype = C.c_void_p
        _lib.orc_load.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int]
        _lib.orc_last_error.restype = C.c_char_p
        for f in ("orc_pass1", "orc_ssaa", "orc_pass1_rows"):
            getattr(_lib, f).restype = C.c_double
        _lib.orc_pass1.argtypes = [C.c_void_p, C.c_void_p]
        _lib.orc_pass1_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        _lib.orc_ssaa.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.orc_sobel.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.orc_fresnel.restype = C.c_float
        _lib.orc_fresnel.argtypes = [C.c_void_p, C.c_void_p, C.c_float]
        _lib.orc_refract.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
        _lib.orc_powf.restype = C.c_float
        _lib.orc_powf.argtypes = [C.c_float, C.c_float]
        _lib.orc_free.argtypes = [C.c_void_p]
        _lib.orc_save_bmp.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_char_p]
        _lib.orc_encode_bmp.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class OracleScene:
    def __init__(self, scene_path, width=-1, height=-1, cwd=ROOT, workers=None):
        self.h = C.c_void_p(lib().orc_load(cwd.encode(), scene_path.encode(), width, height))
        if not self.h:
            raise RuntimeError("orc_load failed: %s" % lib().orc_last_error().decode())
        w, h, no, nl = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        lib().orc_dims(self.h, C.byref(w), C.byref(h), C.byref(no), C.byref(nl))
        self.width, self.height, self.n_objects, self.n_lights = w.value, h.value, no.value, nl.value
        self.last_ms = 0.0
        if workers:
            lib().orc_set_workers(self.h, workers)

    def close(self):
        if self.h:
            lib().orc_free(self.h)
            self.h = None

    def camera(self):
        scale, aspect = C.c_float(), C.c_float()
        m = np.zeros(16, np.float32)
        pos = np.zeros(3, np.float32)
        lib().orc_camera(self.h, C.byref(scale), C.byref(aspect), _p(m), _p(pos))
        return np.float32(scale.value), np.float32(aspect.value), m, pos

    def pass1(self, rows=None):
        fb = np.zeros((self.height, self.width, 3), np.float32)
        if rows is None:
            self.last_ms = lib().orc_pass1(self.h, _p(fb))
        else:
            self.last_ms = lib().orc_pass1_rows(self.h, _p(fb), int(rows[0]), int(rows[1]))
        return fb

    def sobel(self, fb):
        fb = np.ascontiguousarray(fb, np.float32)
        mask = np.zeros((self.height, self.width), np.uint8)
        lib().orc_sobel(self.h, _p(fb), _p(mask))
        return mask

    def ssaa(self, fb, mask=None):
        fb = np.ascontiguousarray(fb.copy())
        if mask is None:
            mask = self.sobel(fb)
        mask = np.ascontiguousarray(mask, np.uint8)
        self.last_ms = lib().orc_ssaa(self.h, _p(fb), _p(mask))
        return fb

    def stats(self, fn):
        lib().orc_set_flag(self.h, b"collectStatistics", 1)
        lib().orc_stats_reset(self.h)
        r = fn()
        out = np.zeros(3, np.int64)
        lib().orc_stats(self.h, _p(out))
        lib().orc_set_flag(self.h, b"collectStatistics", 0)
        return r, out

    def probe(self, rays):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        n = rays.shape[0]
        out = np.zeros((n, 8), np.float32)
        col = np.zeros((n, 3), np.float32)
        lib().orc_probe(self.h, n, _p(rays), _p(out), _p(col))
        return out, col

    def skybox(self, d):
        d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
        out = np.zeros_like(d)
        for i in range(d.shape[0]):
            lib().orc_skybox(self.h, _p(d[i]), _p(out[i]))
        return out

    def illuminate(self, light, pts):
        pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 3)
        out = np.zeros((pts.shape[0], 8), np.float32)
        for i in range(pts.shape[0]):
            lib().orc_illuminate(self.h, light, _p(pts[i]), _p(out[i]))
        return out

    def bvh(self, obj_idx):
        cnt = np.zeros(5, np.int64)
        if lib().orc_bvh_counts(self.h, obj_idx, _p(cnt)) != 0:
            return None
        nn, nl, nr, md, nt = [int(x) for x in cnt]
        d = dict(bounds=np.zeros((nn, 6), np.float32), skip=np.zeros(nn, np.int32),
                 leaf_begin=np.zeros(nn, np.int32), leaf_count=np.zeros(nn, np.int32),
                 refs=np.zeros(nr, np.uint32))
        lib().orc_bvh_dump(self.h, obj_idx, _p(d["bounds"]), _p(d["skip"]), _p(d["leaf_begin"]),
                           _p(d["leaf_count"]), _p(d["refs"]))
        tris = np.zeros((nt, 30), np.float32)
        lib().orc_tris(self.h, obj_idx, _p(tris))
        d.update(tris=tris, n_nodes=nn, n_leaves=nl, n_refs=nr, max_depth=md, n_tris=nt)
        return d


def encode_bmp(fb):
    fb = np.ascontiguousarray(fb, np.float32)
    h, w, _ = fb.shape
    out = np.zeros(54 + 3 * w * h, np.uint8)
    lib().orc_encode_bmp(_p(fb), w, h, _p(out))
    return out.tobytes()


def save_bmp(fb, path):
    fb = np.ascontiguousarray(fb, np.float32)
    h, w, _ = fb.shape
    return lib().orc_save_bmp(_p(fb), w, h, path.encode())


def reflect(d, n):
    d = np.ascontiguousarray(d, np.float32); n = np.ascontiguousarray(n, np.float32)
    out = np.zeros(3, np.float32)
    lib().orc_reflect(_p(d), _p(n), _p(out))
    return out


def refract(d, n, ior):
    d = np.ascontiguousarray(d, np.float32); n = np.ascontiguousarray(n, np.float32)
    out = np.zeros(3, np.float32)
    lib().orc_refract(_p(d), _p(n), C.c_float(ior), _p(out))
    return out


def fresnel(d, n, ior):
    d = np.ascontiguousarray(d, np.float32); n = np.ascontiguousarray(n, np.float32)
    return np.float32(lib().orc_fresnel(_p(d), _p(n), C.c_float(ior)))


def normalize(v):
    v = np.ascontiguousarray(v, np.float32)
    out = np.zeros(3, np.float32)
    lib().orc_normalize(_p(v), _p(out))
    return out


def powf(x, y):
    return np.float32(lib().orc_powf(C.c_float(x), C.c_float(y)))
