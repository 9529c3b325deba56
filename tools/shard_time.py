"""GPU box: what ONE rank of an N-GPU run spends per frame (emulated on one GPU by rendering only that rank's bands).
python tools/shard_time.py [N ...] [--size S] -> per-part pass-1 / SSAA kernel times; the slowest part bounds the N-GPU frame.
(No 8-GPU node is available to the builder: these are projections from one GPU, not measurements of a multi-GPU run.)"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import rendering_amd as RA
from rendering_amd import parallel
args = sys.argv[1:]
S = 4096
if "--size" in args:
    k = args.index("--size"); S = int(args[k + 1]); del args[k:k + 2]
Ns = [int(a) for a in args] or [2, 4, 8]
W = H = S
g = RA.Scene("scenes/cfg2_smooth_250k.scene", W, H)
fb = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
mask = torch.zeros((H, W), dtype=torch.uint8, device="cuda")
for it in range(3):
    parallel.shard_frame(g, fb, mask, 1, 0)
torch.cuda.synchronize()
one = g.last_kernel_ms(0) + g.last_kernel_ms(1) + g.last_kernel_ms(2)
print("%dx%d, 1 GPU: pass 1 %.3f + Sobel %.3f + SSAA %.3f = %.3f ms of kernels" % (W, H, g.last_kernel_ms(0), g.last_kernel_ms(1), g.last_kernel_ms(2), one))
for N in Ns:
    worst = 0
    for part in range(N):
        for it in range(3):      # the third frame uses the cost order of the second
            parallel.shard_frame(g, fb, mask, N, part)
        torch.cuda.synchronize()
        t = g.last_kernel_ms(0) + g.last_kernel_ms(1) + g.last_kernel_ms(2)
        worst = max(worst, t)
    print("N = %d: slowest part %.3f ms of kernels (ideal %.3f) -> projected efficiency %.0f %%" % (N, worst, one / N, 100.0 * one / N / worst))
