"""rendering_amd -- MI355X (gfx950) implementation of the holoskii/Rendering per-pixel ray-trace hot path.

Python is plumbing only: this module binds (ctypes)
  * librtx_hip.so          -- the C ABI of include/rtx.h: HIP kernels + one-time scene upload
  * librendering_host.so   -- the C++17 host side (Scene/Options/Object API, .scene/OBJ/BMP loaders, BVH build)
and moves device pointers of torch tensors across that boundary (torch is used for device memory, streams and
torch.distributed only).  There is NO CPU fallback: without the compiled libraries `load()` raises, and
without a GPU every render call fails loudly with the HIP error.  Nothing here imports oracle/.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB_RTX = os.path.join(HERE, "librtx_hip.so")
LIB_HOST = os.path.join(HERE, "librendering_host.so")

__all__ = ["load", "Scene", "Comm", "RtxError", "device_count", "math_probe", "gather_plan", "exported_symbols"]


class RtxError(RuntimeError):
    pass


class Counters(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("box_tests", C.c_uint64), ("tri_tests", C.c_uint64), ("moot_rays", C.c_uint64)]


_rtx = None
_host = None

# every entry point declared in include/rtx.h (the boundary) and include/rtx_debug.h (probes, diagnostics, tuning hooks)
RTX_SYMBOLS = [
    "rtx_last_error", "rtx_device_count", "rtx_scene_create", "rtx_scene_destroy", "rtx_scene_set_view", "rtx_scene_bytes",
    "rtx_render_pass1", "rtx_sobel", "rtx_render_ssaa", "rtx_render_frame", "rtx_frame_status", "rtx_frame_mode", "rtx_set_frame_mode", "rtx_set_knob", "rtx_cost_grid_read", "rtx_mesh_flatten_probe", "rtx_wide_node_slots", "rtx_source_p_probe", "rtx_quantize_bgr8", "rtx_render_frame_host",
    "rtx_counters_enable", "rtx_counters_reset", "rtx_counters_read", "rtx_last_kernel_ms", "rtx_math_probe",
    "rtx_cast_rays", "rtx_kernel_time_reset", "rtx_kernel_time_stats", "rtx_tile_cost_read", "rtx_set_row_ownership",
    "rtx_bvh_build", "rtx_bvh_info", "rtx_bvh_read", "rtx_bvh_destroy",
    "rtx_vec_probe", "rtx_desc_serialize", "rtx_bvh_build_mode", "rtx_bvh_launches", "rtx_comm_unique_id", "rtx_comm_create", "rtx_comm_info", "rtx_comm_destroy", "rtx_comm_agree", "rtx_gather", "rtx_gather_plan",
]


def load():
    """Loads the native libraries (raises RtxError when they have not been built: run ./build.sh)."""
    global _rtx, _host
    if _rtx is not None:
        return _rtx, _host
    for p in (LIB_RTX, LIB_HOST):
        if not os.path.exists(p):
            raise RtxError("native library %s is missing -- build it with ./build.sh (hipcc, gfx950); "
                           "there is no CPU fallback" % p)
    try:
        # let torch's bundled HIP runtime (same SONAME libamdhip64.so.7) win if torch is going to be used
        import torch  # noqa: F401
    except Exception:
        pass
    rtx = C.CDLL(LIB_RTX, mode=C.RTLD_GLOBAL)
    host = C.CDLL(LIB_HOST, mode=C.RTLD_GLOBAL)
    rtx.rtx_last_error.restype = C.c_char_p
    vp, u32, i32 = C.c_void_p, C.c_uint32, C.c_int
    rtx.rtx_device_count.argtypes = [C.POINTER(C.c_int)]
    rtx.rtx_render_pass1.argtypes = [vp, u32, u32, vp, vp]
    rtx.rtx_sobel.argtypes = [vp, vp, u32, u32, vp, vp]
    rtx.rtx_render_ssaa.argtypes = [vp, vp, u32, u32, vp, vp]
    rtx.rtx_render_frame.argtypes = [vp, u32, u32, vp, vp, vp]
    rtx.rtx_frame_status.argtypes = [vp, C.POINTER(C.c_uint32)]
    rtx.rtx_set_frame_mode.argtypes = [vp, C.c_int]
    rtx.rtx_set_knob.argtypes = [vp, C.c_char_p, C.c_double]
    rtx.rtx_cost_grid_read.argtypes = [vp, vp, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    rtx.rtx_mesh_flatten_probe.argtypes = [vp, C.POINTER(C.c_uint32), vp, vp, C.c_uint32, vp]
    rtx.rtx_source_p_probe.argtypes = [vp, u32, vp, C.c_double, i32, vp]
    rtx.rtx_frame_mode.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_float)]
    rtx.rtx_quantize_bgr8.argtypes = [vp, vp, vp, vp]
    rtx.rtx_render_frame_host.argtypes = [vp, i32, vp]
    rtx.rtx_counters_enable.argtypes = [vp, i32]
    rtx.rtx_counters_reset.argtypes = [vp]
    rtx.rtx_counters_read.argtypes = [vp, C.POINTER(Counters)]
    rtx.rtx_last_kernel_ms.argtypes = [vp, i32, C.POINTER(C.c_float)]
    rtx.rtx_math_probe.argtypes = [i32, i32, u32, vp, vp, vp]
    rtx.rtx_cast_rays.argtypes = [vp, u32, vp, vp, vp]
    rtx.rtx_kernel_time_reset.argtypes = [vp]
    rtx.rtx_kernel_time_stats.argtypes = [vp, i32, C.POINTER(C.c_uint32), C.POINTER(C.c_double)]
    rtx.rtx_tile_cost_read.argtypes = [vp, vp, C.c_size_t]
    rtx.rtx_set_row_ownership.argtypes = [vp, u32, u32, u32, i32]
    rtx.rtx_bvh_build.argtypes = [vp, u32, vp, vp, C.c_int32, i32, C.POINTER(vp)]
    rtx.rtx_bvh_info.argtypes = [vp, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32), C.POINTER(C.c_float)]
    rtx.rtx_bvh_launches.argtypes = [vp, C.POINTER(u32), C.POINTER(C.c_int)]
    rtx.rtx_bvh_read.argtypes = [vp, vp, vp, vp, vp, vp]
    rtx.rtx_bvh_destroy.argtypes = [vp]
    rtx.rtx_bvh_destroy.restype = None
    rtx.rtx_scene_bytes.argtypes = [vp, C.POINTER(C.c_size_t)]
    rtx.rtx_vec_probe.argtypes = [i32, i32, u32, vp, vp, C.c_float, vp]
    rtx.rtx_comm_unique_id.argtypes = [vp]
    rtx.rtx_comm_create.argtypes = [vp, i32, i32, i32, C.POINTER(vp)]
    rtx.rtx_comm_info.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    rtx.rtx_comm_destroy.argtypes = [vp]
    rtx.rtx_comm_agree.argtypes = [vp, C.c_int, C.POINTER(C.c_int), vp]
    rtx.rtx_comm_destroy.restype = None
    rtx.rtx_gather.argtypes = [vp, vp, vp, C.c_size_t, i32, i32, vp]
    rtx.rtx_gather_plan.argtypes = [u32, u32, u32, C.c_size_t, i32, u32, vp, vp, vp, C.POINTER(u32)]
    host.rah_set_ac_build.argtypes = [i32, i32]
    host.rah_set_ac_build.restype = None
    host.rah_set_ac_build_device.argtypes = [i32]
    host.rah_set_ac_build_device.restype = None
    host.rah_bvh_build_info.argtypes = [vp, i32, C.POINTER(C.c_int), C.POINTER(C.c_float)]
    host.rah_last_error.restype = C.c_char_p
    host.rah_scene_load.restype = vp
    host.rah_scene_load.argtypes = [C.c_char_p, C.c_char_p, i32, i32]
    host.rah_scene_free.argtypes = [vp]
    host.rah_scene_dims.argtypes = [vp] + [C.POINTER(C.c_int)] * 4
    host.rah_scene_resize.argtypes = [vp, i32, i32]
    host.rah_scene_set_device.argtypes = [vp, i32]
    host.rah_set_flag.argtypes = [vp, C.c_char_p, i32]
    host.rah_flatten.restype = vp
    host.rah_flatten.argtypes = [vp]
    host.rah_flat_desc.restype = vp
    host.rah_flat_desc.argtypes = [vp]
    host.rah_flat_free.argtypes = [vp]
    host.rah_bvh_counts.argtypes = [vp, i32, vp]
    host.rah_scene_gpu.restype = vp
    host.rah_scene_gpu.argtypes = [vp]
    host.rah_render_host.argtypes = [vp, vp, i32]
    host.rah_save_bmp.argtypes = [vp, vp, C.c_char_p]
    host.rah_bvh_dump.argtypes = [vp, i32, vp, vp, vp, vp, vp]
    host.rah_tris.argtypes = [vp, i32, vp]
    host.rah_camera.argtypes = [vp, vp, vp, vp, vp]
    host.rah_scene_digest.argtypes = [vp, vp, i32]
    host.rah_view_flags.argtypes = [vp]
    host.rah_load_bmp.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), vp, i32]
    _rtx, _host = rtx, host
    return rtx, host


def exported_symbols():
    """(declared, missing) C-ABI symbols of librtx_hip.so -- used by the CPU-side load test."""
    rtx, _ = load()
    missing = [s for s in RTX_SYMBOLS if not hasattr(rtx, s)]
    return list(RTX_SYMBOLS), missing


def _check(rc, what):
    if rc != 0:
        raise RtxError("%s failed (%d): %s" % (what, rc, _rtx.rtx_last_error().decode(errors="replace")))


def device_count():
    rtx, _ = load()
    n = C.c_int(0)
    _check(rtx.rtx_device_count(C.byref(n)), "rtx_device_count")
    return n.value


def _np_ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def set_ac_build(mode, device=0):
    """Where the host loader builds acceleration structures from now on: "host", "device" (rtx_bvh_build) or
    "auto" (the device when one is visible).  Both give the same structure bit for bit."""
    _, host = load()
    host.rah_set_ac_build({"auto": -1, "host": 0, "device": 1}[mode], device)


def bvh_build(tri_pos, root_lo, root_hi, ac_penalty=1, device=0):
    """rtx_bvh_build + rtx_bvh_read: the reference's acceleration structure (objects.cpp:470-526, 633-763) built on
    the GPU.  Returns the dump layout of Scene.bvh() plus `build_ms` (HIP events around the build)."""
    rtx, _ = load()
    pos = np.ascontiguousarray(tri_pos, np.float32).reshape(-1, 9)
    lo = np.ascontiguousarray(root_lo, np.float32); hi = np.ascontiguousarray(root_hi, np.float32)
    b = C.c_void_p()
    _check(rtx.rtx_bvh_build(_np_ptr(pos), pos.shape[0], _np_ptr(lo), _np_ptr(hi), ac_penalty, device, C.byref(b)), "rtx_bvh_build")
    try:
        nn, nr, md, ms = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_float()
        _check(rtx.rtx_bvh_info(b, C.byref(nn), C.byref(nr), C.byref(md), C.byref(ms)), "rtx_bvh_info")
        d = dict(bounds=np.zeros((nn.value, 6), np.float32), skip=np.zeros(nn.value, np.int32),
                 leaf_begin=np.zeros(nn.value, np.int32), leaf_count=np.zeros(nn.value, np.int32),
                 refs=np.zeros(nr.value, np.uint32))
        _check(rtx.rtx_bvh_read(b, _np_ptr(d["bounds"]), _np_ptr(d["skip"]), _np_ptr(d["leaf_begin"]), _np_ptr(d["leaf_count"]),
                                _np_ptr(d["refs"])), "rtx_bvh_read")
        d.update(n_nodes=nn.value, n_refs=nr.value, max_depth=md.value, build_ms=ms.value)
        ln, q = C.c_uint32(), C.c_int()
        _check(rtx.rtx_bvh_launches(b, C.byref(ln), C.byref(q)), "rtx_bvh_launches")
        d.update(launches=ln.value, queued=bool(q.value))
        return d
    finally:
        rtx.rtx_bvh_destroy(b)


def bvh_build_mode(mode):
    """0: rtx_bvh_build in its persistent launches (default); 1: level by level (the fallback of the former; tests compare the two)."""
    rtx, _ = load()
    _check(rtx.rtx_bvh_build_mode(int(mode)), "rtx_bvh_build_mode")


def bvh_build_host(tri_pos, root_lo, root_hi, ac_penalty=1):
    """The HOST builder (rendering_amd/host/src/objects.cpp) on a bare triangle array, in the dump layout of Scene.bvh(); no GPU."""
    _, host = load()
    pos = np.ascontiguousarray(tri_pos, np.float32).reshape(-1, 9)
    lo = np.ascontiguousarray(root_lo, np.float32); hi = np.ascontiguousarray(root_hi, np.float32)
    cnt = np.zeros(4, np.int64)
    host.rah_bvh_from_tris.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p] + [C.c_void_p] * 5
    if host.rah_bvh_from_tris(_np_ptr(pos), pos.shape[0], _np_ptr(lo), _np_ptr(hi), ac_penalty, _np_ptr(cnt), None, None, None, None, None) != 0:
        raise RtxError("rah_bvh_from_tris: %s" % host.rah_last_error().decode(errors="replace"))
    nn, nl, nr, md = [int(x) for x in cnt]
    d = dict(bounds=np.zeros((nn, 6), np.float32), skip=np.zeros(nn, np.int32), leaf_begin=np.zeros(nn, np.int32), leaf_count=np.zeros(nn, np.int32),
             refs=np.zeros(max(nr, 1), np.uint32)[:nr])
    host.rah_bvh_from_tris(_np_ptr(pos), pos.shape[0], _np_ptr(lo), _np_ptr(hi), ac_penalty, _np_ptr(cnt), _np_ptr(d["bounds"]), _np_ptr(d["skip"]),
                           _np_ptr(d["leaf_begin"]), _np_ptr(d["leaf_count"]), _np_ptr(d["refs"]))
    d.update(n_nodes=nn, n_leaves=nl, n_refs=nr, max_depth=md)
    return d


def gather_plan(height, band_height, n_parts, row_bytes, bottom_up=False):
    """rtx_gather_plan: [(owner, byte offset, bytes), ...] -- the transfers rtx_gather executes (host only)."""
    rtx, _ = load()
    n = C.c_uint32(0)
    cap = (height + band_height - 1) // band_height
    owner = np.zeros(cap, np.uint32); off = np.zeros(cap, np.uint64); ln = np.zeros(cap, np.uint64)
    _check(rtx.rtx_gather_plan(height, band_height, n_parts, row_bytes, int(bottom_up), cap, _np_ptr(owner), _np_ptr(off), _np_ptr(ln), C.byref(n)),
           "rtx_gather_plan")
    return [(int(owner[i]), int(off[i]), int(ln[i])) for i in range(n.value)]


class Comm:
    """RCCL communicator behind the C ABI (rtx_comm_*): one rank per GPU.  `exchange(id_bytes_or_None)` must hand rank 0's
    128-byte id to every rank (e.g. a torch.distributed / MPI broadcast, a pipe): it receives the id on rank 0 and None
    elsewhere, and returns the id on every rank."""

    def __init__(self, n_ranks, rank, device, exchange):
        self.rtx, _ = load()
        buf = (C.c_ubyte * 128)()
        if rank == 0:
            _check(self.rtx.rtx_comm_unique_id(buf), "rtx_comm_unique_id")
        raw = exchange(bytes(buf) if rank == 0 else None)
        if len(raw) != 128:
            raise RtxError("communicator id must be 128 bytes")
        self.h = C.c_void_p()
        _check(self.rtx.rtx_comm_create(raw, n_ranks, rank, device, C.byref(self.h)), "rtx_comm_create")
        self.n_ranks, self.rank = n_ranks, rank

    def gather(self, scene, img, bottom_up=False, root=0, stream=None):
        """rtx_gather: every rank's owned rows of the device tensor img (H, ...) end up in rank `root`'s img."""
        row_bytes = img[0].numel() * img.element_size()
        _check(self.rtx.rtx_gather(scene.gpu(), self.h, C.c_void_p(img.data_ptr()), row_bytes, int(bottom_up), root,
                                   Scene._stream_ptr(stream)), "rtx_gather")

    def info(self):
        """rtx_comm_info: (n_ranks, rank) as the RCCL communicator itself reports them (ncclCommCount / ncclCommUserRank)."""
        n, r = C.c_int(0), C.c_int(0)
        _check(self.rtx.rtx_comm_info(self.h, C.byref(n), C.byref(r)), "rtx_comm_info")
        return n.value, r.value

    def agree(self, ok, stream=None):
        """rtx_comm_agree: True iff every rank passed ok=True (called before gather, so that a failed rank does not leave the others waiting)."""
        out = C.c_int(0)
        _check(self.rtx.rtx_comm_agree(self.h, int(bool(ok)), C.byref(out), Scene._stream_ptr(stream)), "rtx_comm_agree")
        return bool(out.value)

    def close(self):
        if self.h:
            self.rtx.rtx_comm_destroy(self.h)
            self.h = C.c_void_p()


class _RtxMesh(C.Structure):
    _fields_ = [("n_nodes", C.c_uint32), ("n_refs", C.c_uint32), ("n_tris", C.c_uint32), ("node_bounds", C.c_void_p), ("node_skip", C.c_void_p),
                ("leaf_begin", C.c_void_p), ("leaf_count", C.c_void_p), ("refs", C.c_void_p), ("tri_pos", C.c_void_p), ("tri_nrm", C.c_void_p),
                ("tri_uv", C.c_void_p), ("tri_tb", C.c_void_p), ("diffuse_w", C.c_uint32), ("diffuse_h", C.c_uint32), ("diffuse_map", C.c_void_p),
                ("normal_w", C.c_uint32), ("normal_h", C.c_uint32), ("normal_map", C.c_void_p), ("specular_w", C.c_uint32), ("specular_h", C.c_uint32),
                ("specular_map", C.c_void_p)]


def mesh_flatten_probe(bvh):
    """Host only: the wide nodes and prune blocks rtx_scene_create derives from a mesh (rtx_mesh_flatten_probe).  bvh = Scene.bvh(obj).
    Returns (wide [n, S, 8] float32 -- link / first as bit patterns in [..., 6:8] --, box records [n, S, 8], plane records [n, S, 8], root record [8]);
    S = rtx_wide_node_slots()."""
    rtx, _ = load()
    bounds = np.ascontiguousarray(bvh["bounds"], np.float32); skip = np.ascontiguousarray(bvh["skip"], np.int32)
    lb = np.ascontiguousarray(bvh["leaf_begin"], np.int32); lc = np.ascontiguousarray(bvh["leaf_count"], np.int32)
    refs = np.ascontiguousarray(bvh["refs"], np.uint32); pos = np.ascontiguousarray(bvh["tris"][:, 0:9], np.float32)
    m = _RtxMesh()
    m.n_nodes, m.n_refs, m.n_tris = len(skip), len(refs), len(pos)
    m.node_bounds, m.node_skip, m.leaf_begin, m.leaf_count = bounds.ctypes.data, skip.ctypes.data, lb.ctypes.data, lc.ctypes.data
    m.refs, m.tri_pos = refs.ctypes.data, pos.ctypes.data
    n = C.c_uint32(0)
    root = np.zeros(8, np.float32)
    _check(rtx.rtx_mesh_flatten_probe(C.byref(m), C.byref(n), None, None, 0, _np_ptr(root)), "rtx_mesh_flatten_probe")
    S = int(rtx.rtx_wide_node_slots())
    wide = np.zeros((n.value, S, 8), np.float32); prune = np.zeros((n.value, 2 * S, 8), np.float32)
    _check(rtx.rtx_mesh_flatten_probe(C.byref(m), C.byref(n), _np_ptr(wide), _np_ptr(prune), n.value, _np_ptr(root)), "rtx_mesh_flatten_probe")
    return wide, prune[:, 0:S], prune[:, S:2 * S], root


def source_p_probe(v0, e1, e2, S, sigma, cam):
    """Host only: P of the source copies of the prune records for triangles (v0, e1, e2: [n, 3] float32), source point S, radius sigma;
    cam: the rays start at S (rtx_source_p_probe; csrc/rtx_source.hip)."""
    rtx, _ = load()
    t = np.ascontiguousarray(np.concatenate([v0, e1, e2], 1), np.float32)
    S = np.ascontiguousarray(S, np.float64)
    out = np.zeros(len(t), np.float32)
    _check(rtx.rtx_source_p_probe(_np_ptr(t), len(t), _np_ptr(S), float(sigma), 1 if cam else 0, _np_ptr(out)), "rtx_source_p_probe")
    return out


def math_probe(op, x, y=None, device=0):
    """Evaluates the device math the parity contract depends on (see rtx_math_probe in include/rtx.h)."""
    rtx, _ = load()
    x = np.ascontiguousarray(x, np.float32)
    out = np.zeros_like(x)
    yp = None
    if y is not None:
        y = np.ascontiguousarray(np.broadcast_to(np.asarray(y, np.float32), x.shape))
        yp = _np_ptr(y)
    _check(rtx.rtx_math_probe(device, op, x.size, _np_ptr(x), yp, _np_ptr(out)), "rtx_math_probe")
    return out


def vec_probe(op, a, b=None, ior=1.0, device=0):
    """rtx_vec_probe: reflect / refract / fresnel / normalize on the device (n x 3 arrays)."""
    rtx, _ = load()
    a = np.ascontiguousarray(a, np.float32).reshape(-1, 3)
    out = np.zeros_like(a)
    bp = None
    if b is not None:
        b = np.ascontiguousarray(b, np.float32).reshape(-1, 3)
        bp = _np_ptr(b)
    _check(rtx.rtx_vec_probe(device, op, a.shape[0], _np_ptr(a), bp, ior, _np_ptr(out)), "rtx_vec_probe")
    return out


class Scene:
    """Host Scene (loaded by the C++ loader) + its uploaded GPU twin.

    Mirrors the reference's Scene surface for the hot path: render() = launchWorkers + launchSSAA
    (scene.cpp:595-606).  `width`/`height` > 0 override the scene file's resolution.
    """

    def __init__(self, scene_path, width=-1, height=-1, device=0, cwd=ROOT):
        self.rtx, self.host = load()
        self.host.rah_set_ac_build_device(device)      # a rank builds its structures on the GPU it renders on
        self.h = C.c_void_p(self.host.rah_scene_load(cwd.encode(), scene_path.encode(), width, height))
        if not self.h:
            raise RtxError("could not load scene %s: %s" % (scene_path, self.host.rah_last_error().decode(errors="replace")))
        self.device = device
        self.host.rah_scene_set_device(self.h, device)
        self._dims()
        self._gpu = None

    def _dims(self):
        w, h, no, nl = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        self.host.rah_scene_dims(self.h, C.byref(w), C.byref(h), C.byref(no), C.byref(nl))
        self.width, self.height, self.n_objects, self.n_lights = w.value, h.value, no.value, nl.value

    def close(self):
        if self.h:
            self.host.rah_scene_free(self.h)
            self.h = None
            self._gpu = None

    def resize(self, width, height):
        self.host.rah_scene_resize(self.h, width, height)
        self._dims()

    def camera_pose(self):
        """(position, rotation in degrees) of the camera (Camera::pos / Camera::rot)."""
        pos, rot = np.zeros(3, np.float32), np.zeros(3, np.float32)
        self.host.rah_camera_get.argtypes = [C.c_void_p] * 3
        self.host.rah_camera_get(self.h, _np_ptr(pos), _np_ptr(rot))
        return pos, rot

    def set_camera(self, pos, rot):
        """Moves the camera; the next render re-applies the view (rtx_scene_set_view: queued on the device, the host does not wait)."""
        pos = np.ascontiguousarray(pos, np.float32); rot = np.ascontiguousarray(rot, np.float32)
        self.host.rah_camera_set.argtypes = [C.c_void_p] * 3
        self.host.rah_camera_set(self.h, _np_ptr(pos), _np_ptr(rot))

    def set_flag(self, name, value):
        self.host.rah_set_flag(self.h, name.encode(), int(value))

    # ---- host-side structures (no GPU needed) -------------------------------------------------
    def camera(self):
        scale, aspect = C.c_float(), C.c_float()
        m = np.zeros(16, np.float32)
        pos = np.zeros(3, np.float32)
        self.host.rah_camera(self.h, C.byref(scale), C.byref(aspect), _np_ptr(m), _np_ptr(pos))
        return np.float32(scale.value), np.float32(aspect.value), m, pos

    def view_flags(self):
        """rtx_view::flags this scene uploads (bit 0 back-face culling, bit 1 skybox)."""
        return self.host.rah_view_flags(self.h)

    def digest(self):
        """Numeric fields of all objects and lights as uploaded (loader tests)."""
        out = np.zeros(4096, np.float32)
        n = self.host.rah_scene_digest(self.h, _np_ptr(out), out.size)
        return out[:n].copy()

    def bvh(self, obj_idx):
        cnt = np.zeros(5, np.int64)
        if self.host.rah_bvh_counts(self.h, obj_idx, _np_ptr(cnt)) != 0:
            return None
        nn, nl, nr, md, nt = [int(x) for x in cnt]
        d = dict(bounds=np.zeros((nn, 6), np.float32), skip=np.zeros(nn, np.int32),
                 leaf_begin=np.zeros(nn, np.int32), leaf_count=np.zeros(nn, np.int32),
                 refs=np.zeros(nr, np.uint32))
        self.host.rah_bvh_dump(self.h, obj_idx, _np_ptr(d["bounds"]), _np_ptr(d["skip"]), _np_ptr(d["leaf_begin"]),
                               _np_ptr(d["leaf_count"]), _np_ptr(d["refs"]))
        tris = np.zeros((nt, 30), np.float32)
        self.host.rah_tris(self.h, obj_idx, _np_ptr(tris))
        d.update(tris=tris, n_nodes=nn, n_leaves=nl, n_refs=nr, max_depth=md, n_tris=nt)
        on, ms = C.c_int(0), C.c_float(0)
        self.host.rah_bvh_build_info(self.h, obj_idx, C.byref(on), C.byref(ms))
        d.update(built_on_device=bool(on.value), build_ms=ms.value)
        return d

    def bvh_build_info(self, obj_idx):
        """(built on the device?, build ms) of object obj_idx's acceleration structure; None for objects without one."""
        on, ms = C.c_int(0), C.c_float(-1)
        if self.host.rah_bvh_build_info(self.h, obj_idx, C.byref(on), C.byref(ms)) != 0:
            return None
        return bool(on.value), ms.value

    # ---- GPU ------------------------------------------------------------------------------------
    def gpu(self):
        """rtx_scene* of the uploaded scene (flatten + upload on first use; re-applies the view after resize)."""
        self._gpu = C.c_void_p(self.host.rah_scene_gpu(self.h))
        if not self._gpu:
            raise RtxError("no GPU scene: %s" % self.host.rah_last_error().decode(errors="replace"))
        return self._gpu

    @staticmethod
    def _stream_ptr(stream):
        if stream is None:
            try:
                import torch
                if torch.cuda.is_available():
                    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
            except Exception:
                pass
            return None
        return C.c_void_p(getattr(stream, "cuda_stream", stream))

    def render_pass1(self, fb, rows=None, stream=None):
        """Scene::launchWorkers on rows [r0,r1) into the device tensor fb (H,W,3 float32, pre-zeroed)."""
        r0, r1 = rows if rows is not None else (0, self.height)
        _check(self.rtx.rtx_render_pass1(self.gpu(), r0, r1, C.c_void_p(fb.data_ptr()), self._stream_ptr(stream)),
               "rtx_render_pass1")

    def sobel(self, fb, mask, rows=None, stream=None):
        r0, r1 = rows if rows is not None else (0, self.height)
        _check(self.rtx.rtx_sobel(self.gpu(), C.c_void_p(fb.data_ptr()), r0, r1, C.c_void_p(mask.data_ptr()),
                                  self._stream_ptr(stream)), "rtx_sobel")

    def render_ssaa(self, mask, fb, rows=None, stream=None):
        r0, r1 = rows if rows is not None else (0, self.height)
        _check(self.rtx.rtx_render_ssaa(self.gpu(), C.c_void_p(mask.data_ptr()), r0, r1, C.c_void_p(fb.data_ptr()),
                                        self._stream_ptr(stream)), "rtx_render_ssaa")

    def render_frame(self, fb, mask, rows=None, stream=None):
        """Pass 1 + Sobel + SSAA of rows [r0,r1) in one launch (rtx_render_frame): same fb and mask as the three calls."""
        r0, r1 = rows if rows is not None else (0, self.height)
        _check(self.rtx.rtx_render_frame(self.gpu(), r0, r1, C.c_void_p(fb.data_ptr()), C.c_void_p(mask.data_ptr()),
                                         self._stream_ptr(stream)), "rtx_render_frame")

    def set_knob(self, name, value):
        """Experiment / test knob of this scene (include/rtx.h, rtx_set_knob); no knob changes a pixel."""
        _check(self.rtx.rtx_set_knob(self.gpu(), name.encode(), float(value)), "rtx_set_knob")

    def cost_grid(self):
        """First-frame cost estimate of the current view: (refs, leaves) per cell of 2 x 2 tiles, arrays grid_h x grid_w."""
        gw, gh = C.c_uint32(0), C.c_uint32(0)
        _check(self.rtx.rtx_cost_grid_read(self.gpu(), None, 0, C.byref(gw), C.byref(gh)), "rtx_cost_grid_read")
        g = np.zeros((gh.value, gw.value, 2), np.uint32)
        _check(self.rtx.rtx_cost_grid_read(self.gpu(), _np_ptr(g), g.size, C.byref(gw), C.byref(gh)), "rtx_cost_grid_read")
        return g[..., 0], g[..., 1]

    def frame_status(self):
        """The host's synchronisation point for render_frame: 0, or error | 0x100 when the single launch gave up and the
        frame was rendered again in three launches (include/rtx.h)."""
        st = C.c_uint32(0)
        _check(self.rtx.rtx_frame_status(self.gpu(), C.byref(st)), "rtx_frame_status")
        return st.value

    def set_frame_mode(self, mode):
        """render_frame: -1 measure and choose (default), 0 always three launches, 1 always one launch."""
        _check(self.rtx.rtx_set_frame_mode(self.gpu(), int(mode)), "rtx_set_frame_mode")

    def frame_mode(self):
        """(mode of the last render_frame: 0 three launches / 1 one launch, measured split ms, measured fused ms; -1 = not yet)."""
        m, a, b = C.c_int(-1), C.c_float(-1), C.c_float(-1)
        _check(self.rtx.rtx_frame_mode(self.gpu(), C.byref(m), C.byref(a), C.byref(b)), "rtx_frame_mode")
        return m.value, a.value, b.value

    def quantize(self, fb, out, stream=None):
        _check(self.rtx.rtx_quantize_bgr8(self.gpu(), C.c_void_p(fb.data_ptr()), C.c_void_p(out.data_ptr()),
                                          self._stream_ptr(stream)), "rtx_quantize_bgr8")

    def render_host(self, ssaa=True):
        """Whole frame into a numpy array through the host-buffer convenience entry (PCIe-inclusive)."""
        fb = np.zeros((self.height, self.width, 3), np.float32)
        _check(self.rtx.rtx_render_frame_host(self.gpu(), int(bool(ssaa)), _np_ptr(fb)), "rtx_render_frame_host")
        return fb

    def cast_rays(self, rays):
        """Render::trace + Render::castRay(depth 0) for n probe rays (n x 6 host array) -> (hits n x 8, colours n x 3)."""
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        n = rays.shape[0]
        hits = np.zeros((n, 8), np.float32)
        col = np.zeros((n, 3), np.float32)
        _check(self.rtx.rtx_cast_rays(self.gpu(), n, _np_ptr(rays), _np_ptr(hits), _np_ptr(col)), "rtx_cast_rays")
        return hits, col

    def counters_enable(self, on=True):
        _check(self.rtx.rtx_counters_enable(self.gpu(), int(on)), "rtx_counters_enable")

    def counters_reset(self):
        _check(self.rtx.rtx_counters_reset(self.gpu()), "rtx_counters_reset")

    def counters(self):
        c = Counters()
        _check(self.rtx.rtx_counters_read(self.gpu(), C.byref(c)), "rtx_counters_read")
        self.moot_rays = int(c.moot_rays)      # shadow rays that cannot influence the pixel (only the instrumented variant traces them)
        return np.array([c.rays, c.box_tests, c.tri_tests], np.int64)

    def scene_bytes(self):
        n = C.c_size_t(0)
        _check(self.rtx.rtx_scene_bytes(self.gpu(), C.byref(n)), "rtx_scene_bytes")
        return n.value

    def last_kernel_ms(self, which=0):
        ms = C.c_float(0)
        _check(self.rtx.rtx_last_kernel_ms(self.gpu(), which, C.byref(ms)), "rtx_last_kernel_ms")
        return ms.value

    def kernel_time_reset(self):
        _check(self.rtx.rtx_kernel_time_reset(self.gpu()), "rtx_kernel_time_reset")

    def kernel_time_stats(self, which=0):
        """(launches, total_ms) of kernel `which` (0 pass 1, 1 sobel, 2 ssaa, 3 whole frame) since kernel_time_reset()."""
        n, ms = C.c_uint32(0), C.c_double(0)
        _check(self.rtx.rtx_kernel_time_stats(self.gpu(), which, C.byref(n), C.byref(ms)), "rtx_kernel_time_stats")
        return n.value, ms.value

    def tile_cost(self):
        """(ceil(H/8), ceil(W/8)) uint32 array: 100 MHz ticks pass 1 spent on each 8x8 tile (rtx_tile_cost_read)."""
        self._dims()
        out = np.zeros(((self.height + 7) // 8, (self.width + 7) // 8), np.uint32)
        _check(self.rtx.rtx_tile_cost_read(self.gpu(), _np_ptr(out), out.size), "rtx_tile_cost_read")
        return out

    def ssaa_item_cost(self):
        """(ceil(H/8), ceil(W/8)) uint32 array: per tile, the slowest SSAA work item of the most recent render_ssaa in
        100 MHz ticks, scaled to a 16-pixel item (0 = the tile had no flagged pixel); the second half of rtx_tile_cost_read."""
        self._dims()
        out = np.zeros((2, (self.height + 7) // 8, (self.width + 7) // 8), np.uint32)
        _check(self.rtx.rtx_tile_cost_read(self.gpu(), _np_ptr(out), out.size), "rtx_tile_cost_read")
        return out[1]

    def set_row_ownership(self, band_height, n_parts, part, halo=True):
        _check(self.rtx.rtx_set_row_ownership(self.gpu(), band_height, n_parts, part, int(halo)), "rtx_set_row_ownership")

    def save_bmp(self, fb, name_no_ext):
        fb = np.ascontiguousarray(fb, np.float32)
        return self.host.rah_save_bmp(self.h, _np_ptr(fb), name_no_ext.encode())
