"""GPU box: fits the first-frame cost estimate (rtx_api.hip estimateCosts: ticks per tile = base + perRef refs + perLeaf leaves of the
2 x 2-tile cell) to the tile costs pass 1 measures, and reports how well the estimate ranks the tiles.
python tools/cost_fit.py [scene W H]..."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import rendering_amd as RA
from rendering_amd import assets
assets.ensure(); assets.ensure(["bumpy_250k.obj"])
cases = [("scenes/cfg2_smooth_250k.scene", 4096, 4096), ("scenes/cfg2_smooth_250k.scene", 1920, 1080), ("scenes/cfg4_textured_1024.scene", 4096, 4096), ("scenes/cfg2_smooth_250k.scene", 8192, 8192)]
X, Y = [], []
for path, W, H in cases:
    g = RA.Scene(path, W, H)
    refs, leaves = g.cost_grid()
    est = g.tile_cost().astype(np.float64)
    fb = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    g.set_frame_mode(0)
    for _ in range(3):
        g.render_pass1(fb)
    torch.cuda.synchronize()
    cost = g.tile_cost().astype(np.float64)
    ty, tx = cost.shape
    r = np.repeat(np.repeat(refs, 2, 0), 2, 1)[:ty, :tx].astype(np.float64); l = np.repeat(np.repeat(leaves, 2, 0), 2, 1)[:ty, :tx].astype(np.float64)
    m = l > 0
    A = np.stack([r[m], l[m], np.ones(m.sum())], 1)
    coef, *_ = np.linalg.lstsq(A, cost[m], rcond=None)
    pred = A @ coef
    # how much of the true top 1 % (by measured cost) does the estimate put into its own top 5 %
    k1 = max(1, int(0.01 * m.sum())); top = np.argsort(-cost[m])[:k1]; top_est = set(np.argsort(-est[m])[:5 * k1].tolist())
    print("%s %dx%d: tiles with leaves %d, measured mean %.0f max %.0f ticks; fit perRef %.2f perLeaf %.1f base %.0f (corr %.3f); current estimate corr %.3f, top-1%% tiles found in its top 5%%: %.0f%%; tiles without leaves: mean %.0f" %
          (os.path.basename(path), W, H, int(m.sum()), cost[m].mean(), cost[m].max(), coef[0], coef[1], coef[2], np.corrcoef(pred, cost[m])[0, 1], np.corrcoef(est[m], cost[m])[0, 1],
           100.0 * len(set(top.tolist()) & top_est) / k1, cost[~m].mean() if (~m).any() else 0))
    X.append(A); Y.append(cost[m])
    g.close()
A = np.concatenate(X); y = np.concatenate(Y)
coef, *_ = np.linalg.lstsq(A, y, rcond=None)
print("all cases together: perRef %.2f perLeaf %.1f base %.0f" % tuple(coef))
