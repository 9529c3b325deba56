"""GPU box: acceleration-structure build time, host builder vs rtx_bvh_build."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import rendering_amd as RA

# The HIP runtime and the code object are initialised once per process whoever touches the GPU first (about 0.1 s); a
# render needs them anyway, so they are paid here before anything is timed (round 1 charged them to the first device build).
RA.bvh_build(np.array([[0, 0, 0, 1, 0, 0, 0, 1, 0]], np.float32), [0, 0, 0], [1, 1, 1], 1)

for name in sys.argv[1:] or ["cfg2_smooth_250k", "cfg2_smooth_25k", "cfg4_textured_1024"]:
    path = "scenes/%s.scene" % name
    t_host_load = t_dev_load = 1e9
    for _ in range(3):
        RA.set_ac_build("host")
        t0 = time.perf_counter(); gh = RA.Scene(path, 64, 64); t_host_load = min(t_host_load, time.perf_counter() - t0)
        RA.set_ac_build("device")
        t0 = time.perf_counter(); gd = RA.Scene(path, 64, 64); t_dev_load = min(t_dev_load, time.perf_counter() - t0)
    for oi in range(gh.n_objects):
        h = gh.bvh(oi)
        if h is None:
            continue
        pen = 3 if "cfg4" in name else 1
        best = 1e9; wall = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            d = RA.bvh_build(h["tris"][:, :9], h["bounds"][0, :3], h["bounds"][0, 3:], pen)
            wall = min(wall, time.perf_counter() - t0)
            best = min(best, d["build_ms"])
        same = all(np.array_equal(h[k].view(np.uint32), d[k].view(np.uint32)) for k in ("bounds", "skip", "leaf_begin", "leaf_count", "refs"))
        print("%s obj %d: %d tris -> %d nodes, %d refs, depth %d | scene load host-build %.1f ms, device-build %.1f ms | "
              "rtx_bvh_build: GPU %.2f ms, wall incl. H2D/D2H %.2f ms | identical %s"
              % (name, oi, h["n_tris"], d["n_nodes"], d["n_refs"], d["max_depth"], t_host_load * 1e3, t_dev_load * 1e3, best, wall * 1e3, same))
RA.set_ac_build("auto")
