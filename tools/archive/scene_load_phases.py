"""GPU box: where the scene load goes -- host load (.scene / OBJ parse + acceleration structure, built on the device) and upload (flatten + rtx_scene_create + view), for a
first and a second instance of the same scene in one process (the first pays one-off costs: code object load, first allocations).  python tools/scene_load_phases.py [scene W H]"""
import os, sys, time, ctypes as C
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import rendering_amd as RA
scene = sys.argv[1] if len(sys.argv) > 1 else "scenes/cfg2_smooth_250k.scene"
W = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
H = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
torch.zeros(1, device="cuda"); torch.cuda.synchronize()
for k in range(3):
    t0 = time.perf_counter()
    g = RA.Scene(scene, W, H)
    t1 = time.perf_counter()
    g.gpu(); torch.cuda.synchronize()
    t2 = time.perf_counter()
    fb = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda"); mask = torch.zeros((H, W), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    g.render_frame(fb, mask); torch.cuda.synchronize()
    t4 = time.perf_counter()
    print("instance %d: host load %.1f ms (device BVH build %s), flatten + upload + view %.1f ms, first frame %.2f ms" % (k, (t1 - t0) * 1e3, g.bvh_build_info(1), (t2 - t1) * 1e3, (t4 - t3) * 1e3))
    g.close()
