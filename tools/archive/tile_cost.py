"""GPU box: where pass 1 spends its time.  python tools/tile_cost.py [scene] [W] [H] -> gpurun_out/tile_cost.npy + a coarse map."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import rendering_amd as RA

scene = sys.argv[1] if len(sys.argv) > 1 else "scenes/cfg2_smooth_250k.scene"
W = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
H = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
g = RA.Scene(scene, W, H)
fb = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
g.render_pass1(fb); g.render_pass1(fb)
torch.cuda.synchronize()
c = g.tile_cost().astype(np.float64) * 1e-5      # ms
os.makedirs("gpurun_out", exist_ok=True)
np.save("gpurun_out/tile_cost.npy", c.astype(np.float32))
print("tiles", c.shape, "sum of tile times %.1f ms" % c.sum(), "max %.3f ms" % c.max(), "median %.4f ms" % np.median(c))
s = np.sort(c.ravel())[::-1]
cs = np.cumsum(s) / s.sum()
for f in (0.001, 0.01, 0.05, 0.1, 0.25, 0.5):
    print("  top %5.1f%% of tiles hold %5.1f%% of the time" % (f * 100, 100 * cs[int(f * len(s)) - 1]))
B = 32
ty, tx = c.shape
m = c[:ty // B * B, :tx // B * B].reshape(ty // B, B, tx // B, B).sum((1, 3))
print("coarse map (%% of total per %dx%d-tile block):" % (B, B))
for row in m:
    print(" ".join("%4.1f" % (100 * v / c.sum()) for v in row))
