"""GPU box: why the first frame of a view of the area-light scene takes twice its warm frame: the first-frame estimate of the tile costs beside
what the frame kernel then measures (which tiles it would split, where the time is): python tools/area_first_frame_probe.py [scene W H]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import rendering_amd as RA
scene = sys.argv[1] if len(sys.argv) > 1 else "scenes/area_light.scene"
W = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
H = int(sys.argv[3]) if len(sys.argv) > 3 else 1080


def timed(f):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); f(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)


def limits(c, waves=4096, percent=100, floor=2000):
    s4 = max(int(c.sum() / waves * percent / 100), floor)
    return s4, 4 * s4


def run(mode, label):
    g = RA.Scene(scene, W, H)
    if mode is not None: g.set_frame_mode(mode)
    fb = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda"); mask = torch.zeros((H, W), dtype=torch.uint8, device="cuda")
    g.gpu(); torch.cuda.synchronize()
    est = g.tile_cost().astype(np.float64)
    first = timed(lambda: g.render_frame(fb, mask)); m1 = g.frame_mode()
    after1 = g.tile_cost().astype(np.float64)
    later = [timed(lambda: g.render_frame(fb, mask)) for _ in range(8)]
    meas = g.tile_cost().astype(np.float64)
    print("%s: first frame %.3f ms (mode %s), the eight after it %s" % (label, first, m1, " ".join("%.3f" % x for x in later)))
    return est, after1, meas


est, after1, meas = run(None, "as shipped")
run(0, "three launches forced")
run(1, "one launch forced")
ty, tx = np.mgrid[0:est.shape[0], 0:est.shape[1]]
for name, c in (("estimate", est), ("measured by the first frame", after1), ("measured warm", meas)):
    s4, s16 = limits(c)
    print("%-28s sum %.1f ms, mean %.0f ticks, max %.0f; split limits %d / %d ticks: %d tiles in 4, %d in 16" % (name, c.sum() * 1e-5, c.mean(), c.max(), s4, s16, ((c > s4) & (c <= s16)).sum(), (c > s16).sum()))
ok = (est > 0) & (after1 > 0)
print("correlation of log(estimate) with log(first frame's costs): %.3f" % np.corrcoef(np.log(est[ok]), np.log(after1[ok]))[0, 1])
idx = np.argsort(after1.ravel())[::-1][:24]
print("slowest tiles of the first frame (tx,ty): measured / estimated ticks:", " ".join("(%d,%d) %d/%d" % (i % est.shape[1], i // est.shape[1], after1.ravel()[i], est.ravel()[i]) for i in idx))
# the share of the first frame's wave time by estimate decile
order = np.argsort(est.ravel()); dec = np.array_split(order, 10)
print("share of the measured time by decile of the estimate:", " ".join("%.1f%%" % (100 * after1.ravel()[d].sum() / after1.sum()) for d in dec))
h, e = np.histogram(np.log2(np.maximum(after1.ravel(), 1)), bins=16, range=(8, 24)); print("log2(ticks) histogram of the first frame's costs 8..24:", h.tolist())
h, e = np.histogram(np.log2(np.maximum(est.ravel(), 1)), bins=16, range=(8, 24)); print("log2(ticks) histogram of the estimate 8..24:", h.tolist())
