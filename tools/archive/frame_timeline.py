"""GPU box, RTX_DBG build: what the frame kernel (rtx_render_frame) is doing when.
RTX_DBG_TIMELINE=/tmp/tl.bin python tools/frame_timeline.py [scene W H]"""
import os as _os
_os.environ.setdefault("RTX_ALLOW_ENV_KNOBS", "1")      # (the product ignores RTX_* environment knobs without it)
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import rendering_amd as RA
scene = sys.argv[1] if len(sys.argv) > 1 else "scenes/cfg2_smooth_250k.scene"
W = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
H = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
parts = int(sys.argv[4]) if len(sys.argv) > 4 else 1
part = int(sys.argv[5]) if len(sys.argv) > 5 else 0
path = os.environ.setdefault("RTX_DBG_TIMELINE", "/tmp/tl.bin")
g = RA.Scene(scene, W, H)
fb = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda"); mask = torch.zeros((H, W), dtype=torch.uint8, device="cuda")
g.set_frame_mode(1)
g.set_row_ownership(64 if parts > 1 else 0, parts, part, True)
for it in range(4):
    if it >= 2:
        os.environ["RTX_DBG_TIMELINE"] = path      # (the dump of frame 2 empties the buffer; the one of frame 3 is read)
    else:
        os.environ.pop("RTX_DBG_TIMELINE", None)
    g.render_frame(fb, mask); g.frame_status()
print("frame kernel %.3f ms" % g.last_kernel_ms(3))
tl = np.fromfile(path, dtype=np.uint64).reshape(-1, 3)
start = tl[:, 0].astype(np.int64); dur = tl[:, 1].astype(np.int64); wave = (tl[:, 2] >> np.uint64(48)).astype(np.int64); kind = ((tl[:, 2] >> np.uint64(32)) & np.uint64(0xffff)).astype(np.int64); item = (tl[:, 2] & np.uint64(0xffffffff)).astype(np.int64)
t0 = start.min(); start = (start - t0) * 1e-5; dur = dur * 1e-5; end = start + dur       # ms (100 MHz)
span = end.max()
print("items %d (pass-1 tiles %d, SSAA items %d), span %.3f ms; busy wave-ms: pass 1 %.1f, SSAA %.1f" % (len(tl), (kind == 1).sum(), (kind == 2).sum(), span, dur[kind == 1].sum(), dur[kind == 2].sum()))
for k, name in ((1, "pass-1 tile"), (2, "SSAA item")):
    m = kind == k
    if not m.any(): continue
    order = np.argsort(-dur[m])[:8]
    print("slowest %s: " % name + ", ".join("%.3f ms (start %.3f)" % (dur[m][i], start[m][i]) for i in order))
    print("  last %s ends at %.3f ms; starts by tenth of the span: %s" % (name, end[m].max(), np.histogram(start[m], bins=10, range=(0, span))[0].tolist()))
last = np.argsort(-end)[:10]
print("last to finish: " + ", ".join("%s %.3f..%.3f" % ("P1" if kind[i] == 1 else "SS", start[i], end[i]) for i in last))
# the SSAA items of the slowest tiles: when could they start (their tile's end) and when did they
m2 = kind == 2
if m2.any():
    tiles_x = (W + 7) // 8
    t_of = item[m2] >> 8
    p1_end = {}
    for i in np.nonzero(kind == 1)[0]:
        tx, ty = item[i] & 0xffff, item[i] >> 16
        p1_end[ty * tiles_x + tx] = end[i]
    lag = np.array([start[m2][j] - p1_end.get(int(t_of[j]), 0.0) for j in range(m2.sum())])
    print("SSAA start minus the end of its own tile's pass 1: median %.3f ms, 90%% %.3f, max %.3f" % (np.median(lag), np.percentile(lag, 90), lag.max()))
    worst = np.argsort(-end[m2])[:5]
    print("last SSAA items: " + ", ".join("tile pass-1 end %.3f, item %.3f..%.3f (%s)" % (p1_end.get(int(t_of[j]), 0.0), start[m2][j], end[m2][j], ("16px", "4px", "1px", "?")[item[m2][j] & 3]) for j in worst))

# time a wave spends between two of its work items (queues, counters, Sobel)
order = np.lexsort((start, wave))
w_s, s_s, e_s = wave[order], start[order], end[order]
same = w_s[1:] == w_s[:-1]
gap = (s_s[1:] - e_s[:-1])[same]
print("between two items of a wave: median %.1f us, 90%% %.1f us, max %.1f us; per wave in all %.3f ms of the %.3f ms span (busy %.3f ms)"
      % (np.median(gap) * 1e3, np.percentile(gap, 90) * 1e3, gap.max() * 1e3, gap.sum() / max(1, len(np.unique(wave))), span, dur.sum() / max(1, len(np.unique(wave)))))
