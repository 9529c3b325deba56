#!/usr/bin/env python3
"""Build container only (needs /root/reference and oracle/_ref): the reference's OWN models through the reference's OWN loader
and builder (objects.cpp:177-381, 470-526, 633-763) -> tests/golden/ref_models.npz.

For every OBJ under /root/reference/input/objects (quads / n-gons, faces without vn / vt, the `max = numeric_limits<float>::min()`
quirk at objects.cpp:231, the clipped root box of a rotated mesh) and a few placements / penalties, the harness loads a one-mesh
scene with the reference itself and the file keeps
    * pos_<case>    : n_tris x 9 fp32 -- a, b, c of every triangle as the reference's loader produced them (INPUT of the builder tests
                      that run without the OBJ: the GPU box has no /root/reference)
    * root_<case>   : 6 fp32 -- bounds of the root node (objects.cpp:328-330)
    * meta json     : counts (tris, nodes, leaves, refs, depth), penalty, sha1 of the full 30-float triangle records, of the node
                      bounds, skip links, leaf begin / count arrays and of the leaf references.
The OBJ files themselves are not copied into the repository.  tests/test_ref_models.py checks the host loader + builder (here) and
rtx_bvh_build (GPU box) against it."""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_OBJ = "/root/reference/input/objects"

# (case name, obj, pos, size, rot, ac_penalty)
CASES = [
    ("bunny", "bunny.obj", "0,0,-3", "2,2,2", "0,0,0", 1),
    ("bunny_rot_p3", "bunny.obj", "0.2,-0.1,-2.5", "1.5,1.5,1.5", "10,200,-30", 3),
    ("cow", "cow.obj", "0,0,-3", "2,2,2", "0,30,0", 1),
    ("teapot", "teapot.obj", "0,-0.5,-4", "3,3,3", "0,0,0", 1),
    ("teapot_p2", "teapot.obj", "1,0,-4", "2,1,2", "-20,45,5", 2),
    ("sphere", "sphere.obj", "0,0,-3", "2,2,2", "0,0,0", 1),
    ("shotgun", "shotgun.obj", "-0.1,0,-0.6", "2,2,2", "0,100,0", 3),
    ("icosahedron", "icosahedron.obj", "0,0,-2", "1,1,1", "0,0,0", 1),
    ("floor", "floor.obj", "0,-1,-3", "4,4,4", "0,0,0", 1),
]

SCENE = """[options]
width=64
height=64
ac_penalty=%d
image_name=output/ref_model

[light]
type=point
position=0,2,0
color=1,1,1
intensity=1.0

[object]
type=mesh
pos=%s
size=%s
color=1,1,1
rot=%s
material=diffuse
name=%s

[end]
"""

CHILD = r"""
import sys, json, hashlib, numpy as np
sys.path.insert(0, %r)
from tools import ref_harness as R
r = R.RefScene(sys.argv[1], 64, 64, cwd='/')
b = r.bvh(0)
np.savez(sys.argv[2], **{k: v for k, v in b.items() if isinstance(v, np.ndarray)})
json.dump({k: int(v) for k, v in b.items() if not isinstance(v, np.ndarray)}, open(sys.argv[2] + '.json', 'w'))
"""


def sha(a):
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()


def scene_text(case):
    name, obj, pos, size, rot, pen = case
    return SCENE % (pen, pos, size, rot, os.path.join(REF_OBJ, obj))


def main():
    if not os.path.isdir(REF_OBJ):
        raise SystemExit("needs /root/reference (build container)")
    out = {}
    meta = {}
    with tempfile.TemporaryDirectory() as tmp:
        for case in CASES:
            name = case[0]
            sp = os.path.join(tmp, name + ".scene")
            open(sp, "w").write(scene_text(case))
            dump = os.path.join(tmp, name + ".npz")
            # one scene per process: the reference keeps process-global option flags
            subprocess.run([sys.executable, "-c", CHILD % ROOT, sp, dump], check=True, cwd=ROOT)
            d = np.load(dump)
            cnt = json.load(open(dump + ".json"))
            tris = d["tris"]
            out["pos_" + name] = tris[:, :9].copy()
            out["root_" + name] = d["bounds"][0].copy()
            meta[name] = dict(obj=case[1], pos=case[2], size=case[3], rot=case[4], ac_penalty=case[5], n_tris=cnt["n_tris"], n_nodes=cnt["n_nodes"],
                              n_leaves=cnt["n_leaves"], n_refs=cnt["n_refs"], max_depth=cnt["max_depth"], sha_tris=sha(tris), sha_bounds=sha(d["bounds"]),
                              sha_skip=sha(d["skip"]), sha_leaf_begin=sha(d["leaf_begin"]), sha_leaf_count=sha(d["leaf_count"]), sha_refs=sha(d["refs"]))
            print(name, {k: meta[name][k] for k in ("n_tris", "n_nodes", "n_leaves", "n_refs", "max_depth")})
    out["meta"] = np.frombuffer(json.dumps(meta, sort_keys=True).encode(), np.uint8)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ref_models.npz"), **out)
    print("wrote tests/golden/ref_models.npz")


if __name__ == "__main__":
    main()
