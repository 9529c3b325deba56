"""GPU box, RTX_DBG build: the work items of the SSAA launch (three-launch path) -- when each started, how long it took.
python tools/ssaa_timeline.py [scene W H [parts part]]"""
import os as _os
_os.environ.setdefault("RTX_ALLOW_ENV_KNOBS", "1")      # (the product ignores RTX_* environment knobs without it)
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import rendering_amd as RA
scene = sys.argv[1] if len(sys.argv) > 1 else "scenes/cfg2_smooth_250k.scene"
W = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
H = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
parts = int(sys.argv[4]) if len(sys.argv) > 4 else 1
part = int(sys.argv[5]) if len(sys.argv) > 5 else 0
path = "/tmp/ssaa_tl.bin"
g = RA.Scene(scene, W, H)
fb = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda"); mask = torch.zeros((H, W), dtype=torch.uint8, device="cuda")
g.set_frame_mode(0)
g.set_row_ownership(64 if parts > 1 else 0, parts, part, True)
for it in range(4):
    if it >= 2: os.environ["RTX_DBG_TIMELINE"] = path      # (the dump of frame 2 empties the buffer; the one of frame 3 is read)
    else: os.environ.pop("RTX_DBG_TIMELINE", None)
    g.render_frame(fb, mask); g.frame_status()
print("pass 1 %.3f ms, SSAA stage %.3f ms (list kernels included), frame %.3f ms" % (g.last_kernel_ms(0), g.last_kernel_ms(2), g.last_kernel_ms(3)))
tl = np.fromfile(path, dtype=np.uint64).reshape(-1, 3)
start = tl[:, 0].astype(np.int64); dur = tl[:, 1].astype(np.int64); wave = (tl[:, 2] >> np.uint64(48)).astype(np.int64); pxy = (tl[:, 2] & np.uint64(0xffffffff)).astype(np.int64)
t0 = start.min(); start = (start - t0) * 1e-5; dur = dur * 1e-5; end = start + dur
print("items %d on %d waves, span %.3f ms, busy %.1f wave-ms (%.3f ms if spread over %d waves)" % (len(tl), len(np.unique(wave)), end.max(), dur.sum(), dur.sum() / 4096, 4096))
print("duration percentiles (ms): " + ", ".join("%d%% %.3f" % (q, np.percentile(dur, q)) for q in (50, 90, 99, 99.9, 100)))
print("starts by tenth of the span: %s" % np.histogram(start, bins=10, range=(0, end.max()))[0].tolist())
for i in np.argsort(-end)[:12]:
    print("  ends %.3f: start %.3f, %.3f ms, first pixel (%d, %d), wave %d" % (end[i], start[i], dur[i], pxy[i] & 0xffff, pxy[i] >> 16, wave[i]))
per = np.bincount(wave)
print("items per wave: max %d, mean %.2f" % (per.max(), per[per > 0].mean()))
