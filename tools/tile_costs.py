"""GPU box: the slowest pass-1 tiles of a view and the distribution of the tile costs: python tools/tile_costs.py [scene] [W] [H] [n]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import rendering_amd as RA
scene = sys.argv[1] if len(sys.argv) > 1 else "scenes/cfg2_smooth_250k.scene"
W = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
H = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
n = int(sys.argv[4]) if len(sys.argv) > 4 else 12
g = RA.Scene(scene, W, H)
fb = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
g.render_pass1(fb); g.render_pass1(fb)
torch.cuda.synchronize()
c = g.tile_cost().astype(np.float64) * 1e-5      # ms
print("pass1 %.3f ms; tile cost sum %.1f ms, mean %.4f, median %.4f, 99%% %.4f, 99.9%% %.4f, max %.3f" % (g.last_kernel_ms(0), c.sum(), c.mean(), np.median(c), np.quantile(c, 0.99), np.quantile(c, 0.999), c.max()))
idx = np.argsort(c.ravel())[::-1][:n]
print("slowest:", " ".join("(%d,%d)=%.3f" % (i % c.shape[1], i // c.shape[1], c.ravel()[i]) for i in idx))
h, e = np.histogram(np.log10(np.maximum(c.ravel(), 1e-5)), bins=12, range=(-3, 1))
print("log10(ms) histogram -3..1:", h.tolist())
# share of the wave-time by region (headline scene: the mesh's disc has a radius of ~1255 px around the centre at 4096^2; the floor lies below the horizon y = H / 2)
ty, tx = np.mgrid[0:c.shape[0], 0:c.shape[1]]
r = np.hypot((tx + 0.5) * 8 - W / 2, (ty + 0.5) * 8 - H / 2) * 4096.0 / W
mesh = r < 1420; sky = (~mesh) & ((ty + 1) * 8 <= H / 2); floor = (~mesh) & ~sky
for name, m in (("mesh disc", mesh), ("floor", floor), ("sky", sky)):
    print("%-9s tiles %7d (%.1f%%), wave-time %8.1f ms (%.1f%%), mean %.4f ms" % (name, m.sum(), 100.0 * m.mean(), c[m].sum(), 100.0 * c[m].sum() / c.sum(), c[m].mean()))
fl = c[floor]
print("floor tiles: median %.4f, 90%% %.4f, 99%% %.4f, max %.3f ms; by tile row (mean ms):" % (np.median(fl), np.quantile(fl, 0.9), np.quantile(fl, 0.99), fl.max()),
      " ".join("%d:%.3f" % (y, c[y][floor[y]].mean()) for y in range(c.shape[0] // 2, c.shape[0], 16) if floor[y].any()))
