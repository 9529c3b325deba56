"""GPU box: what a NEW VIEW costs the host (rtx_scene_set_view queues its work and returns) and the device: python tools/new_view_probe.py [scene W H]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import rendering_amd as RA
scene = sys.argv[1] if len(sys.argv) > 1 else "scenes/cfg2_smooth_250k.scene"
W = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
H = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
g = RA.Scene(scene, W, H)
fb = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda"); mask = torch.zeros((H, W), dtype=torch.uint8, device="cuda")
for _ in range(4): g.render_frame(fb, mask)
torch.cuda.synchronize()
pos, rot = g.camera_pose()
host, dev, stages, warm = [], [], [], []
for k in range(8):
    for _ in range(2): g.render_frame(fb, mask)
    t0 = time.perf_counter()
    g.set_camera(pos + np.float32([0.01 * (k + 1), 0.005 * k, 0]), rot + np.float32([0, 0.5 * (k + 1), 0]))
    g.gpu()
    host.append((time.perf_counter() - t0) * 1e3)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.render_frame(fb, mask); e1.record(); torch.cuda.synchronize()
    dev.append(e0.elapsed_time(e1))
    stages.append((g.last_kernel_ms(0), g.last_kernel_ms(1), g.last_kernel_ms(2)))
    g.render_frame(fb, mask); torch.cuda.synchronize()
    warm.append((g.last_kernel_ms(0), g.last_kernel_ms(1), g.last_kernel_ms(2)))
    assert g.frame_status() == 0
print("new view, host ms per rtx_scene_set_view:", " ".join("%.3f" % x for x in host))
print("first frame of each new view, ms (HIP events):", " ".join("%.3f" % x for x in dev))
print("   its stages (pass 1, Sobel, SSAA; three-launch frames only):", " ".join("%.3f/%.3f/%.3f" % x for x in stages))
print("   the second frame of the view:", " ".join("%.3f/%.3f/%.3f" % x for x in warm))
t0 = time.perf_counter(); g.resize(W, H); g.gpu(); print("same view again: %.3f ms" % ((time.perf_counter() - t0) * 1e3))
