"""Loader and builder parity on the REFERENCE'S OWN models (/root/reference/input/objects/{bunny,cow,teapot,sphere,shotgun,
icosahedron,floor}.obj: quads / n-gons, faces without vn / vt, the numeric_limits<float>::min() quirk of objects.cpp:231, the clipped
root box of a rotated mesh) -- VERDICT r4 "missing" item 2.  The golden file tests/golden/ref_models.npz was written by
tools/make_golden_ref_models.py from the reference itself (its loader objects.cpp:177-381, its builder objects.cpp:470-526, 633-763,
through oracle/_ref/libref_harness.so); it keeps the triangles the reference's loader produced and digests of its acceleration
structure, never the OBJ files.

  * where /root/reference exists (the build container): THIS repo's host loader reads the same OBJ files and must produce the same
    30-float triangle records and the same structure, bit for bit; so must the oracle's own loader / builder;
  * everywhere (no OBJ needed): the host builder on the golden triangles == the reference's structure;
  * on the GPU box: rtx_bvh_build on the golden triangles == the reference's structure."""
import hashlib
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_OBJ = "/root/reference/input/objects"
GOLD = np.load(os.path.join(ROOT, "tests", "golden", "ref_models.npz"))
META = json.loads(GOLD["meta"].tobytes().decode())
CASES = sorted(META)


def sha(a):
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()


def check_structure(name, d):
    m = META[name]
    assert (d["n_nodes"], d["n_refs"], d["max_depth"]) == (m["n_nodes"], m["n_refs"], m["max_depth"]), (name, d["n_nodes"], d["n_refs"], d["max_depth"], m)
    assert sha(d["bounds"]) == m["sha_bounds"], name + ": node bounds"
    assert sha(d["skip"]) == m["sha_skip"], name + ": skip links"
    assert sha(d["leaf_begin"]) == m["sha_leaf_begin"] and sha(d["leaf_count"]) == m["sha_leaf_count"], name + ": leaves"
    assert sha(d["refs"]) == m["sha_refs"], name + ": leaf references (the reference's visiting order)"


@pytest.mark.parametrize("name", CASES)
def test_host_builder_on_the_references_triangles(ra, name):
    d = ra.bvh_build_host(GOLD["pos_" + name], GOLD["root_" + name][:3], GOLD["root_" + name][3:], META[name]["ac_penalty"])
    check_structure(name, d)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_device_builder_on_the_references_triangles(ra, name):
    d = ra.bvh_build(GOLD["pos_" + name], GOLD["root_" + name][:3], GOLD["root_" + name][3:], META[name]["ac_penalty"])
    check_structure(name, d)


def _scene_file(tmp_path, name):
    from tools import make_golden_ref_models as G
    case = [c for c in G.CASES if c[0] == name][0]
    p = tmp_path / (name + ".scene")
    p.write_text(G.scene_text(case))
    return str(p)


@pytest.mark.skipif(not os.path.isdir(REF_OBJ), reason="no /root/reference here (the OBJ files are not copied into the repository)")
@pytest.mark.parametrize("name", CASES)
def test_host_loader_on_the_references_obj_files(ra, tmp_path, name):
    ra.set_ac_build("host")
    try:
        s = ra.Scene(_scene_file(tmp_path, name), 64, 64, cwd="/")
        d = s.bvh(0)
        s.close()
    finally:
        ra.set_ac_build("auto")
    m = META[name]
    assert d["n_tris"] == m["n_tris"]
    assert d["tris"][:, :9].tobytes() == GOLD["pos_" + name].tobytes(), name + ": vertex positions"
    assert sha(d["tris"]) == m["sha_tris"], name + ": normals / uvs / tangents of the 30-float records"
    assert d["bounds"][0].tobytes() == GOLD["root_" + name].tobytes(), name + ": root box"
    check_structure(name, d)


@pytest.mark.skipif(not os.path.isdir(REF_OBJ), reason="no /root/reference here")
@pytest.mark.parametrize("name", CASES)
def test_oracle_loader_on_the_references_obj_files(oracle, tmp_path, name):
    o = oracle.OracleScene(_scene_file(tmp_path, name), 64, 64, cwd="/")
    d = o.bvh(0)
    m = META[name]
    assert sha(d["tris"]) == m["sha_tris"], name
    check_structure(name, d)
