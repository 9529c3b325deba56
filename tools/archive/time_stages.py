"""GPU box: times of the three stages of a frame (HIP events of the C ABI): python tools/time_stages.py [scene] [W] [H] [reps]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import rendering_amd as RA
scene = sys.argv[1] if len(sys.argv) > 1 else "scenes/cfg2_smooth_250k.scene"
W = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
H = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 14
g = RA.Scene(scene, W, H)
fb = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
mask = torch.zeros((H, W), dtype=torch.uint8, device="cuda")
t = []
for rep in range(reps):
    g.render_pass1(fb); g.sobel(fb, mask); g.render_ssaa(mask, fb)
    torch.cuda.synchronize()
    t.append((g.last_kernel_ms(0), g.last_kernel_ms(1), g.last_kernel_ms(2)))
t = np.array(t[reps // 2:])
print("%s %dx%d, last %d of %d frames: pass1 min %.3f median %.3f ms, sobel %.3f, ssaa min %.3f median %.3f; sum of medians %.3f" % (
    os.path.basename(scene), W, H, len(t), reps, t[:, 0].min(), np.median(t[:, 0]), np.median(t[:, 1]), t[:, 2].min(), np.median(t[:, 2]), np.median(t, 0).sum()))
