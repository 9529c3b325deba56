"""GPU box: what ONE rank of an N-GPU run spends per frame (emulated on one GPU by rendering only that rank's bands).
python tools/shard_time.py [N] -> per-part pass-1 / SSAA kernel times; the slowest part bounds the N-GPU frame."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import rendering_amd as RA
from rendering_amd import parallel
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
W = H = 4096
g = RA.Scene("scenes/cfg2_smooth_250k.scene", W, H)
fb = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
mask = torch.zeros((H, W), dtype=torch.uint8, device="cuda")
worst = 0
for part in range(N):
    for it in range(3):      # the third frame uses the cost order of the second
        parallel.shard_frame(g, fb, mask, N, part)
    torch.cuda.synchronize()
    p1, ss = g.last_kernel_ms(0), g.last_kernel_ms(2)
    worst = max(worst, p1 + ss)
    print("part %d/%d: pass 1 %.3f ms, SSAA %.3f ms" % (part, N, p1, ss))
print("slowest part %.3f ms  (1 GPU: ~15.4 ms of kernels -> ideal %.2f ms)" % (worst, 15.4 / N))
