"""GPU box, one-off: rtx_bvh_build (two persistent launches over a queue of nodes, csrc/rtx_bvh.hip bvhq) against the HOST builder (rah_bvh_from_tris) on random triangle
sets -- soups, clustered points, long slivers, many coincident triangles, flat sets, tiny and large counts, penalties 1-6 -- bit for bit.  python tools/bvh_fuzz.py [first seed] [n]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import rendering_amd as RA
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
bad = 0; queued = 0
for seed in range(first, first + n):
    r = np.random.default_rng(seed)
    kind = seed % 6
    nt = int(r.choice([1, 2, 7, 63, 64, 65, 500, 1023, 1025, 5000, 20000, 60000, 150000]))
    ext = float(r.choice([0.05, 0.5, 2.0, 8.0]))
    if kind == 0:      # soup of small triangles
        c = r.uniform(-ext, ext, (nt, 1, 3)); tri = c + r.normal(0, ext * 0.02, (nt, 3, 3))
    elif kind == 1:    # clusters
        k = r.integers(1, 6); cent = r.uniform(-ext, ext, (k, 3)); c = cent[r.integers(0, k, nt)][:, None, :] + r.normal(0, ext * 0.05, (nt, 1, 3)); tri = c + r.normal(0, ext * 0.01, (nt, 3, 3))
    elif kind == 2:    # long slivers through the whole box
        a = r.uniform(-ext, ext, (nt, 3)); b = r.uniform(-ext, ext, (nt, 3)); tri = np.stack([a, b, a + r.normal(0, 1e-3, (nt, 3))], 1)
    elif kind == 3:    # many coincident triangles
        base = r.uniform(-ext, ext, (max(nt // 50, 1), 3, 3)); tri = base[r.integers(0, base.shape[0], nt)]
    elif kind == 4:    # flat: all in a plane (an axis of zero extent)
        c = r.uniform(-ext, ext, (nt, 1, 3)); tri = c + r.normal(0, ext * 0.03, (nt, 3, 3)); tri[:, :, int(r.integers(0, 3))] = 0.25
    else:              # a bumpy sheet (neighbours share vertices, like a mesh)
        m = int(np.ceil(np.sqrt(nt / 2))) + 1; u, v = np.meshgrid(np.linspace(-ext, ext, m), np.linspace(-ext, ext, m)); z = 0.2 * ext * np.sin(3 * u / ext) * np.cos(2 * v / ext)
        P = np.stack([u, z, v], -1); a = P[:-1, :-1]; b = P[:-1, 1:]; cc = P[1:, 1:]; d = P[1:, :-1]
        tri = np.concatenate([np.stack([a, b, cc], 2).reshape(-1, 3, 3), np.stack([a, cc, d], 2).reshape(-1, 3, 3)])[:nt]
    tri = np.ascontiguousarray(tri, np.float32).reshape(-1, 9)
    lo = tri.reshape(-1, 3).min(0) - np.float32(r.choice([0, 1e-3, 0.1])); hi = tri.reshape(-1, 3).max(0) + np.float32(r.choice([0, 1e-3, 0.1]))
    pen = int(r.integers(1, 7))
    h = RA.bvh_build_host(tri, lo, hi, pen)
    d = RA.bvh_build(tri, lo, hi, pen)
    queued += d["queued"]
    same = all(h[k].tobytes() == d[k].tobytes() for k in ("bounds", "skip", "leaf_begin", "leaf_count", "refs")) and h["max_depth"] == d["max_depth"]
    if not same:
        bad += 1
        print("MISMATCH seed %d kind %d tris %d penalty %d: nodes %d / %d refs %d / %d queued %s" % (seed, kind, tri.shape[0], pen, h["n_nodes"], d["n_nodes"], h["n_refs"], d["n_refs"], d["queued"]))
print("seeds %d..%d: %d mismatching builds, %d of %d by the persistent launches; sources %s" % (first, first + n - 1, bad, queued, n, __import__("tools.srchash", fromlist=["x"]).source_hash()))
