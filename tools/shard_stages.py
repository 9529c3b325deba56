"""GPU box: per-stage time of every part of an N-way sharded frame (emulated on one GPU), to see whether the loss against
the ideal 1/N is imbalance between parts (max >> mean) or overhead common to all parts (mean >> whole/N).
python tools/shard_stages.py [--size S] [--bands 32,64,128] [--mode 0|1] [N ...]"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import rendering_amd as RA
from rendering_amd import parallel
args = sys.argv[1:]
S = 4096
bands = [None]
if "--size" in args:
    k = args.index("--size"); S = int(args[k + 1]); del args[k:k + 2]
if "--bands" in args:
    k = args.index("--bands"); bands = [int(b) for b in args[k + 1].split(",")]; del args[k:k + 2]
MODE = -1      # --mode 0 / 1: force three launches / one launch (default: rtx_render_frame chooses per part)
if "--mode" in args:
    k = args.index("--mode"); MODE = int(args[k + 1]); del args[k:k + 2]
Ns = [int(a) for a in args] or [8]
W = H = S
g = RA.Scene("scenes/cfg2_smooth_250k.scene", W, H)
fb = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
mask = torch.zeros((H, W), dtype=torch.uint8, device="cuda")
g.set_frame_mode(MODE)


def stages(parts, part, band):
    best = None
    for it in range(8):
        parallel.shard_frame(g, fb, mask, parts, part, band=band)
        torch.cuda.synchronize()
        if it >= 5:
            t = []
            for k in range(4):
                try: t.append(g.last_kernel_ms(k))
                except RA.RtxError: t.append(0.0)      # (a frame in one launch records no stages)
            if best is None or t[3] < best[3]:
                best = t
    return best


one = stages(1, 0, None)
print("%dx%d whole: pass1 %.3f sobel %.3f ssaa %.3f frame %.3f ms" % (W, H, *one))
for N in Ns:
    for band in bands:
        bh = band or parallel.band_height(H, N)
        rows = []
        for part in range(N):
            rows.append(stages(N, part, bh))
        mean = [sum(r[k] for r in rows) / N for k in range(4)]
        mx = [max(r[k] for r in rows) for k in range(4)]
        print("N=%d band=%d: frame max %.3f mean %.3f ideal %.3f | pass1 max %.3f mean %.3f ideal %.3f | sobel mean %.3f | ssaa max %.3f mean %.3f ideal %.3f"
              % (N, bh, mx[3], mean[3], one[3] / N, mx[0], mean[0], one[0] / N, mean[1], mx[2], mean[2], one[2] / N))
        print("   parts frame ms: " + " ".join("%.3f" % r[3] for r in rows))
