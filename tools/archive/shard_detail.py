"""GPU box: per-part kernel times of an N-way sharded frame in the three-launch mode.  python tools/shard_detail.py SIZE N"""
import os, sys, torch
sys.path.insert(0, ".")
import rendering_amd as RA
from rendering_amd import parallel
S = int(sys.argv[1]); N = int(sys.argv[2])
g = RA.Scene("scenes/cfg2_smooth_250k.scene", S, S)
g.set_frame_mode(0)
fb = torch.zeros((S, S, 3), dtype=torch.float32, device="cuda"); mask = torch.zeros((S, S), dtype=torch.uint8, device="cuda")
for part in range(N):
    for it in range(3): parallel.shard_frame(g, fb, mask, N, part)
    torch.cuda.synchronize()
    print("part %d: pass1 %.3f sobel %.3f ssaa %.3f frame %.3f" % (part, g.last_kernel_ms(0), g.last_kernel_ms(1), g.last_kernel_ms(2), g.last_kernel_ms(3)))
