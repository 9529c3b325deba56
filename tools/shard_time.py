"""GPU box: what ONE rank of an N-GPU run spends per frame (emulated on one GPU by rendering only that rank's bands).
python tools/shard_time.py [N ...] [--size S] -> per-part frame time (rtx_render_frame); the slowest part bounds the N-GPU frame.
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
BAND = None      # (parallel.band_height unless --band)
if "--band" in args:
    k = args.index("--band"); BAND = int(args[k + 1]); del args[k:k + 2]
Ns = [int(a) for a in args] or [2, 4, 8]
W = H = S
g = RA.Scene("scenes/cfg2_smooth_250k.scene", W, H)
fb = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
mask = torch.zeros((H, W), dtype=torch.uint8, device="cuda")
def settled(parts, part):
    """ms of one frame of this part (events around rtx_render_frame), after it has settled on one launch or three."""
    best = 1e9
    for it in range(10):
        parallel.shard_frame(g, fb, mask, parts, part, band=BAND)
        torch.cuda.synchronize()
        if it >= 7:
            best = min(best, g.last_kernel_ms(3))
    mode, a, b = g.frame_mode()
    return best, mode, a, b


one, mode, a, b = settled(1, 0)
print("%dx%d%s, 1 GPU: %.3f ms per frame (%s; measured three launches %.3f ms, one launch %.3f ms)" % (W, H, ", bands of %d rows" % BAND if BAND else "", one, ("three launches", "one launch")[mode], a, b))
for N in Ns:
    worst = 0; modes = ""
    for part in range(N):
        t, mode, a, b = settled(N, part)
        modes += "31"[mode]
        worst = max(worst, t)
    print("N = %d (%d-row bands): slowest part %.3f ms per frame (ideal %.3f; launches per part: %s) -> projected efficiency %.0f %%" % (N, BAND or parallel.band_height(H, N), worst, one / N, modes, 100.0 * one / N / worst))
