"""GPU box (research): candidate first-frame cost estimators against the tile costs pass 1 measures.  The splat of
rtxCostSplatKernel restated in numpy with per-leaf weights: refs, refs x facing class, refs / leaves per depth layer ...
python tools/research/cost_estimators.py [W H]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
os.chdir(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import rendering_amd as RA
from rendering_amd import assets
assets.ensure(["bumpy_250k.obj"])
W = int(sys.argv[1]) if len(sys.argv) > 1 else 1920
H = int(sys.argv[2]) if len(sys.argv) > 2 else 1080
g = RA.Scene("scenes/cfg2_smooth_250k.scene", W, H)
b = g.bvh(1)
fb = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
for _ in range(3):
    g.render_pass1(fb)
torch.cuda.synchronize()
cost = g.tile_cost().astype(np.float64)
ty, tx = cost.shape
scale, aspect, M, pos = g.camera(); M = M.reshape(4, 4)
tris = b["tris"][:, 0:9].astype(np.float64)
lc = b["leaf_count"]; lbeg = b["leaf_begin"]; refs = b["refs"]
leaves = np.nonzero(lc > 0)[0]
gw, gh = (tx + 1) // 2, (ty + 1) // 2
feats = {k: np.zeros((gh, gw)) for k in ("refs", "leaves", "front", "graze", "back", "frontl", "grazel")}
for i in leaves:
    r = refs[lbeg[i]:lbeg[i] + lc[i]]
    T = tris[r]
    V = T.reshape(-1, 3)
    lo, hi = V.min(0), V.max(0)
    n = np.cross(T[:, 3:6] - T[:, 0:3], T[:, 6:9] - T[:, 0:3])
    nm = n.sum(0); nn = np.linalg.norm(nm)
    c = 0.5 * (lo + hi) - pos
    view = c / np.linalg.norm(c)
    nl = np.linalg.norm(n, axis=1) + 1e-300
    cosv = -(n @ view) / nl                        # > 0: faces the camera
    corners = np.array([[lo[0] if k & 1 == 0 else hi[0], lo[1] if k & 2 == 0 else hi[1], lo[2] if k & 4 == 0 else hi[2]] for k in range(8)]) - pos
    s = corners @ M[:3, :3].T
    if (s[:, 2] > -1e-4).any():
        continue
    xp = s[:, 0] / -s[:, 2]; yp = s[:, 1] / -s[:, 2]
    px = (xp / (scale * aspect) + 1) * 0.5 * W - 1; py = (-yp / scale + 1) * 0.5 * H - 1
    cx0, cx1 = max(0, int(np.floor(px.min() / 16))), min(gw - 1, int(np.floor(px.max() / 16)))
    cy0, cy1 = max(0, int(np.floor(py.min() / 16))), min(gh - 1, int(np.floor(py.max() / 16)))
    if cx1 < cx0 or cy1 < cy0:
        continue
    sl = (slice(cy0, cy1 + 1), slice(cx0, cx1 + 1))
    feats["refs"][sl] += len(r); feats["leaves"][sl] += 1
    nf = int((cosv > 0.25).sum()); ng = int((np.abs(cosv) <= 0.25).sum()); nb = len(r) - nf - ng
    feats["front"][sl] += nf; feats["graze"][sl] += ng; feats["back"][sl] += nb
    feats["frontl"][sl] += nf > 0; feats["grazel"][sl] += ng > 0
up = lambda a: np.repeat(np.repeat(a, 2, 0), 2, 1)[:ty, :tx]
F = {k: up(v) for k, v in feats.items()}
m = F["leaves"] > 0
y = cost[m]
k1 = max(1, int(0.01 * m.sum())); top = set(np.argsort(-y)[:k1].tolist())
def report(name, cols):
    A = np.stack([F[c][m] for c in cols] + [np.ones(m.sum())], 1)
    coef, *_ = np.linalg.lstsq(A, y, rcond=None)
    pred = A @ coef
    rec = len(top & set(np.argsort(-pred)[:5 * k1].tolist())) / k1
    print("%-40s corr %.3f  top-1%% in top-5%% %.0f%%  coef %s" % (name, np.corrcoef(pred, y)[0, 1], 100 * rec, np.round(coef, 2)))
print("%dx%d: %d tiles with leaves, mean cost %.0f" % (W, H, m.sum(), y.mean()))
report("refs + leaves (today)", ["refs", "leaves"])
report("front + graze + back", ["front", "graze", "back"])
report("front + graze + back + leaves", ["front", "graze", "back", "leaves"])
report("graze + grazel + front + frontl", ["graze", "grazel", "front", "frontl"])
report("graze only", ["graze"])
