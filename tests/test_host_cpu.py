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
    # the drop-in boundary (rtx.h) + the diagnostics / probes the tests and tools use (rtx_debug.h)
    hdr = open(os.path.join(ROOT, "include", "rtx.h")).read()
    boundary = set(re.findall(r"\b(rtx_[a-z0-9_]+)\s*\(", hdr))
    assert len(boundary) <= 32, "probes and tuning hooks belong in include/rtx_debug.h"
    declared = boundary | set(re.findall(r"\b(rtx_[a-z0-9_]+)\s*\(", open(os.path.join(ROOT, "include", "rtx_debug.h")).read()))
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


def test_legacy_dialect_equals_current_dialect(ra):
    """Loader hardening: the reference's own input/smooth_shading.scene is written in an older
    one-line-per-entity dialect its current loader rejects.  The host loader reads it; the legacy transliteration
    of cfg2 must flatten to exactly the same scene as the current-dialect file."""
    a = ra.Scene("scenes/legacy_smooth_4k.scene", 96, 64)
    b = ra.Scene("scenes/cfg2_smooth_4k.scene", 96, 64)
    assert (a.n_objects, a.n_lights) == (b.n_objects, b.n_lights) == (2, 3)
    assert np.array_equal(a.digest().view(np.uint32), b.digest().view(np.uint32))
    for x, y in zip(a.camera(), b.camera()):
        assert np.array_equal(np.asarray(x), np.asarray(y))
    da, db = a.bvh(1), b.bvh(1)
    for k in db:
        assert np.array_equal(da[k], db[k]) if isinstance(db[k], np.ndarray) else da[k] == db[k]


def test_bmp_loader_hardening(ra, tmp_path):
    """Row padding (width % 4 != 0), 32 bpp, pixel-data offset and top-down files load to the same texels as the
    plain 24-bpp file the reference can read (util.cpp:78-113)."""
    import struct
    import ctypes as C
    _, host = ra.load()
    host.rah_load_bmp.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p, C.c_int]
    rng = np.random.default_rng(3)

    def load(path):
        w, h = C.c_int(), C.c_int()
        buf = np.zeros(1 << 16, np.uint8)
        n = host.rah_load_bmp(str(path).encode(), C.byref(w), C.byref(h), buf.ctypes.data_as(C.c_void_p), buf.size)
        return buf[:n].reshape(h.value, w.value, 3).copy()

    def write(path, rgb, bpp=24, top_down=False, extra=0):
        h, w, _ = rgb.shape
        rows = rgb if top_down else rgb[::-1]
        body = b""
        for r in rows:
            px = np.concatenate([r[:, ::-1], np.full((w, 1), 255, np.uint8)], 1) if bpp == 32 else r[:, ::-1]
            line = px.astype(np.uint8).tobytes()
            body += line + b"\0" * ((-len(line)) % 4)
        off = 54 + extra
        hdr = b"BM" + struct.pack("<IHHI", off + len(body), 0, 0, off)
        hdr += struct.pack("<IiiHHIIiiII", 40, w, -h if top_down else h, 1, bpp, 0, len(body), 2835, 2835, 0, 0)
        open(path, "wb").write(hdr + b"\x55" * extra + body)

    img = rng.integers(0, 256, (6, 8, 3), dtype=np.uint8)           # row 0 = top
    write(tmp_path / "plain.bmp", img)
    ref = load(tmp_path / "plain.bmp")
    assert np.array_equal(ref, img[::-1])                            # rows stay bottom-up, channels RGB
    write(tmp_path / "b32.bmp", img, bpp=32); assert np.array_equal(load(tmp_path / "b32.bmp"), ref)
    write(tmp_path / "td.bmp", img, top_down=True); assert np.array_equal(load(tmp_path / "td.bmp"), ref)
    write(tmp_path / "off.bmp", img, extra=84); assert np.array_equal(load(tmp_path / "off.bmp"), ref)
    odd = rng.integers(0, 256, (5, 7, 3), dtype=np.uint8)            # 7*3 = 21 bytes per row -> 3 bytes of padding
    write(tmp_path / "odd.bmp", odd); assert np.array_equal(load(tmp_path / "odd.bmp"), odd[::-1])


def test_errors_are_reported_not_fatal(ra, tmp_path):
    """Recoverable host errors (missing scene, malformed BMP header, no GPU) come back as RtxError through the C API
    instead of ending the process with the reference's LOG_ERROR() exit."""
    import ctypes as C
    import struct
    import torch
    with pytest.raises(ra.RtxError):
        ra.Scene("scenes/does_not_exist.scene")
    _, host = ra.load()
    w, h = C.c_int(), C.c_int()
    for hdr_w, hdr_h in ((1 << 30, 1 << 30), (16, -(1 << 31)), (0, 4), (70000, 4)):
        p = tmp_path / "bad.bmp"
        p.write_bytes(b"BM" + struct.pack("<IHHI", 54, 0, 0, 54) + struct.pack("<IiiHHIIiiII", 40, hdr_w, hdr_h, 1, 24, 0, 0, 2835, 2835, 0, 0))
        assert host.rah_load_bmp(str(p).encode(), C.byref(w), C.byref(h), None, 0) == -1
        assert b"BMP" in host.rah_last_error()
    if not torch.cuda.is_available():
        s = ra.Scene("scenes/cfg1_simple_shapes.scene", 64, 64)
        with pytest.raises(ra.RtxError):
            s.gpu()


def test_scene_flags_are_per_scene(ra):
    """Loading a second scene does not change the switches an earlier one uploads (the reference keeps them in
    process-global options::; the C API pins them per Scene)."""
    a = ra.Scene("scenes/cfg3_reflective_refractive.scene", 64, 64)       # sets the skybox switch
    assert a.view_flags() == 3
    b = ra.Scene("scenes/cfg1_simple_shapes.scene", 64, 64)
    assert b.view_flags() == 1
    assert a.view_flags() == 3
    b.set_flag("useBackfaceCulling", 0)
    assert b.view_flags() == 0 and a.view_flags() == 3
    import os
    assert os.getcwd() == os.path.dirname(os.path.dirname(os.path.abspath(__file__)))       # the loader restored the cwd


@pytest.mark.parametrize("scene,obj", [("scenes/cfg2_smooth_4k.scene", 1), ("scenes/cfg4_textured_256.scene", 0), ("scenes/coincident.scene", 1)])
def test_prune_records_enclose_what_they_stand_for(ra, scene, obj):
    """The prune blocks of the wide walk (DESIGN_HISTORY.md 3.1c; rtx_mesh_flatten_probe, host only): for every slot of every wide node
    the box record encloses all vertices of all triangles referenced below it and P bounds their |e1|_1 |e2|_1, the plane
    record encloses their scaled normals and plane offsets; the records of a slot enclose those of the wide node below it;
    the wide nodes hold exactly the reference's boxes (log2 S levels apart, S = rtx_wide_node_slots()) and reach every leaf reference once."""
    from rendering_amd import assets
    assets.ensure()
    g = ra.Scene(scene, 64, 48)
    b = g.bvh(obj)
    wide, box, plane, root = ra.mesh_flatten_probe(b)
    assert len(wide) > 0
    S = wide.shape[1]
    assert S in (4, 8, 16)
    tris = b["tris"][:, 0:9].astype(np.float64)
    A, B, C_ = tris[:, 0:3], tris[:, 3:6], tris[:, 6:9]
    e1 = (tris[:, 3:6].astype(np.float32) - tris[:, 0:3].astype(np.float32)).astype(np.float64)      # the fp32 differences of the exact test
    e2 = (tris[:, 6:9].astype(np.float32) - tris[:, 0:3].astype(np.float32)).astype(np.float64)
    s1, s2 = np.abs(e1).sum(1), np.abs(e2).sum(1)
    link = wide[..., 6].view(np.int32); first = wide[..., 7].view(np.int32)
    refs = b["refs"]
    seen = np.zeros(len(refs), np.int32)

    def check_slot(w, k):
        """-> (lo, hi, P, qlo, qhi, wlo, whi) actually spanned below slot k of wide node w (None: nothing)."""
        l = int(link[w, k])
        if l == 0:
            return None
        if l < 0:
            n, f = ~l, int(first[w, k])
            seen[f:f + n] += 1
            r = refs[f:f + n]
            if n == 0:
                return None
            V = np.concatenate([A[r], A[r] + e1[r], A[r] + e2[r]])
            ok = (s1[r] > 0) & (s2[r] > 0)
            q = np.cross(e2[r], e1[r])[ok] / (s1[r] * s2[r])[ok][:, None]
            wv = (A[r][ok] * q).sum(1)
            span = (V.min(0), V.max(0), (s1[r] * s2[r]).max(), q.min(0) if len(q) else None, q.max(0) if len(q) else None, wv.min() if len(q) else None, wv.max() if len(q) else None)
        else:
            parts = [p for p in (check_slot(l - 1, kk) for kk in range(S)) if p is not None]
            if not parts:
                return None
            qs = [p for p in parts if p[3] is not None]
            span = (np.min([p[0] for p in parts], 0), np.max([p[1] for p in parts], 0), max(p[2] for p in parts),
                    np.min([p[3] for p in qs], 0) if qs else None, np.max([p[4] for p in qs], 0) if qs else None,
                    min(p[5] for p in qs) if qs else None, max(p[6] for p in qs) if qs else None)
        c, P, h = box[w, k, 0:3].astype(np.float64), float(box[w, k, 3]), box[w, k, 4:7].astype(np.float64)
        assert (c - h <= span[0]).all() and (c + h >= span[1]).all(), "box record of slot %d of wide node %d" % (k, w)
        assert P >= span[2], "P of slot %d of wide node %d" % (k, w)
        qc, wlo, qr, whi = plane[w, k, 0:3].astype(np.float64), float(plane[w, k, 3]), plane[w, k, 4:7].astype(np.float64), float(plane[w, k, 7])
        if np.isfinite(wlo) and span[3] is not None:      # (a record without a plane bound is q = 0 +- 0, offsets [-inf, +inf]: never rejected)
            assert (qc - qr <= span[3]).all() and (qc + qr >= span[4]).all() and wlo <= span[5] and whi >= span[6], "plane record of slot %d of wide node %d" % (k, w)
            assert (np.abs(qc) + qr <= 1.0 + 1e-5).all()      # |q|_inf <= 1: what planeAlive's error terms assume
        return span

    spans = [p for p in (check_slot(0, k) for k in range(S)) if p is not None]
    assert (seen == 1).all(), "every leaf reference is reached through exactly one slot"
    lo = np.min([p[0] for p in spans], 0); hi = np.max([p[1] for p in spans], 0)
    assert (root[0:3] - root[4:7] <= lo).all() and (root[0:3] + root[4:7] >= hi).all() and root[3] >= max(p[2] for p in spans)
    # the slots are the reference's own boxes: every slot box is one of the node boxes of the binary tree
    nb = {tuple(x) for x in b["bounds"][:, [0, 3, 1, 4, 2, 5]].astype(np.float32).tolist()}
    for w in range(len(wide)):
        for k in range(S):
            if link[w, k] != 0:
                assert tuple(wide[w, k, 0:6].tolist()) in nb


def test_flatten_rejects_a_right_child_outside_the_array(ra):
    """an inner node whose right child index equals n_nodes (a leaf first child at n_nodes - 1) passed the skip-index check
    and made the bottom-up passes of flattenMesh read past the arrays; rtx_mesh_flatten_probe feeds caller arrays straight in."""
    bvh = dict(bounds=np.array([[0, 0, 0, 1, 1, 1], [0, 0, 0, 1, 1, 1], [0, 0, 0, 1, 1, 1]], np.float32),
               skip=np.array([3, 3, 0], np.int32),            # node 1: inner, skip 3 = n_nodes; its first child (node 2) is a leaf -> right child = 3
               leaf_begin=np.array([0, 0, 0], np.int32), leaf_count=np.array([-1, -1, 1], np.int32),
               refs=np.array([0], np.uint32), tris=np.zeros((1, 30), np.float32))
    with pytest.raises(ra.RtxError):
        ra.mesh_flatten_probe(bvh)


@pytest.mark.parametrize("depth,wide_expected", [(12, True), (29, True), (45, False)])
def test_a_tree_deeper_than_the_walks_stack_takes_the_binary_form(ra, depth, wide_expected):
    """The wide walk's stack in LDS holds kWideSlots - 1 entries per wide level + 1 (rtx_device.h: 76 entries for eight slots = ten wide levels = thirty
    binary ones); flattenMesh must not hand out wide nodes for a deeper tree (rtx_scene_create then walks it in the binary form).  A chain: every inner
    node has a leaf as its first child and the rest of the tree as its second."""
    n = 2 * depth + 1                                   # inner nodes at 0, 2, 4, ...; leaves at 1, 3, ... and the last node
    leaf_count = np.full(n, -1, np.int32); leaf_count[1::2] = 1; leaf_count[n - 1] = 1
    skip = np.zeros(n, np.int32); skip[leaf_count < 0] = n
    leaves = np.nonzero(leaf_count >= 0)[0]
    leaf_begin = np.zeros(n, np.int32); leaf_begin[leaves] = np.arange(len(leaves))
    tris = np.zeros((len(leaves), 30), np.float32)
    tris[:, 0:9] = np.array([0.1, 0.1, 0.1, 0.9, 0.1, 0.1, 0.1, 0.9, 0.1], np.float32)
    bvh = dict(bounds=np.tile(np.array([0, 0, 0, 1, 1, 1], np.float32), (n, 1)), skip=skip, leaf_begin=leaf_begin, leaf_count=leaf_count,
               refs=np.arange(len(leaves), dtype=np.uint32), tris=tris)
    wide, box, plane, root = ra.mesh_flatten_probe(bvh)
    assert (len(wide) > 0) == wide_expected
    if wide_expected:
        link = wide[..., 6].view(np.int32)
        assert (link < 0).sum() == len(leaves)          # every leaf sits in exactly one slot


@pytest.mark.parametrize("name", ["r6_ref_bunny", "r6_ref_cow", "r6_ref_teapot", "r6_ref_sphere"])
def test_round6_workload_scenes_load_alike(ra, oracle, name):
    """The round-6 bench workloads built from the reference's own models (rendering_amd/assets.py, ref_model_obj: the triangles the reference's loader produced,
    tests/golden/ref_models.npz, written back as an OBJ): the host's loader + builder and the oracle's (pinned to the reference: tests/test_oracle_vs_reference.py)
    produce the same triangles and the same acceleration structure, bit for bit."""
    from rendering_amd import assets
    assets.ensure()
    s = ra.Scene("scenes/%s.scene" % name, 64, 64)
    o = oracle.OracleScene("scenes/%s.scene" % name, 64, 64)
    a, b = s.bvh(1), o.bvh(1)
    assert a is not None and b is not None and a["n_tris"] == b["n_tris"] > 900
    for k in ("n_nodes", "n_leaves", "n_refs", "max_depth"):
        assert a[k] == b[k], k
    for k in ("bounds", "skip", "leaf_begin", "leaf_count", "refs", "tris"):
        assert np.array_equal(np.ascontiguousarray(a[k]).view(np.uint8), np.ascontiguousarray(b[k]).view(np.uint8)), k


def test_knot_generator_is_a_closed_outward_tube():
    """The second 250k-triangle mesh (assets.knot_obj) at a small size: 2 nu nv faces, every vertex used, and every face's geometric normal on the side of its
    vertex normals (the reference culls by winding: det > 0 is the front, objects.cpp:75-77)."""
    from rendering_amd import assets
    txt = assets.knot_obj(nu=60, nv=12).decode().split("\n")
    v = np.array([[float(t) for t in l.split()[1:]] for l in txt if l.startswith("v ")])
    vn = np.array([[float(t) for t in l.split()[1:]] for l in txt if l.startswith("vn ")])
    f = np.array([[int(t.split("//")[0]) - 1 for t in l.split()[1:]] for l in txt if l.startswith("f ")])
    assert len(v) == 60 * 12 == len(vn) and len(f) == 2 * 60 * 12 and set(f.ravel()) == set(range(len(v)))
    # the reference's front face: det = e1 . (dir x e2) > 0 for a ray coming from outside, i.e. (e2 x e1) points against the outward normal
    m = np.cross(v[f[:, 2]] - v[f[:, 0]], v[f[:, 1]] - v[f[:, 0]])
    outward = vn[f[:, 0]] + vn[f[:, 1]] + vn[f[:, 2]]
    assert ((m * outward).sum(1) < 0).all()
