"""GPU box: raw material for the first-frame cost estimate -- per tile the estimate's inputs (references / leaves of its 2 x 2-tile cell) and the measured pass-1 cost.
python tools/cost_dump.py out.npz [scene W H]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import rendering_amd as RA
out = sys.argv[1]
scene = sys.argv[2] if len(sys.argv) > 2 else "scenes/cfg2_smooth_250k.scene"
W = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
H = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
g = RA.Scene(scene, W, H)
refs, leaves = g.cost_grid()
est = g.tile_cost().copy()
fb = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
for _ in range(3): g.render_pass1(fb)
torch.cuda.synchronize()
np.savez_compressed(out, refs=refs, leaves=leaves, est=est, cost=g.tile_cost())
