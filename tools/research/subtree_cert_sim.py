"""CPU research tool: how many node visits / leaf headers would certificates at INNER nodes (subtree normal interval +
tight AABB) save?  Simulates the stackless wave walk for sampled 8x8 waves of the headline frame."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import rendering_amd as RA
from oracle import oracle as O

W = H = 4096
NT = int(sys.argv[1]) if len(sys.argv) > 1 else 40
RA.set_ac_build("host")
g = RA.Scene("scenes/cfg2_smooth_250k.scene", 64, 64)
d = g.bvh(1)
lc = d["leaf_count"]; lb = d["leaf_begin"]; refs = d["refs"]; bounds = d["bounds"].astype(np.float64); skip = d["skip"]
nN = len(skip)
tr = d["tris"][:, :9].reshape(-1, 3, 3).astype(np.float64)
mvec = np.cross(tr[:, 2] - tr[:, 0], tr[:, 1] - tr[:, 0])
tlo = tr.min(1); thi = tr.max(1)
# subtree aggregates, bottom-up in reverse pre-order
mlo = np.full((nN, 3), np.inf); mhi = np.full((nN, 3), -np.inf); blo = np.full((nN, 3), np.inf); bhi = np.full((nN, 3), -np.inf); cnt = np.zeros(nN, np.int64)
for i in range(nN - 1, -1, -1):
    if lc[i] >= 0:
        ids = refs[lb[i]:lb[i] + lc[i]]
        if len(ids):
            mlo[i] = mvec[ids].min(0); mhi[i] = mvec[ids].max(0); blo[i] = tlo[ids].min(0); bhi[i] = thi[ids].max(0)
        cnt[i] = len(ids)
    else:
        l = i + 1; r = skip[l]
        mlo[i] = np.minimum(mlo[l], mlo[r]); mhi[i] = np.maximum(mhi[l], mhi[r]); blo[i] = np.minimum(blo[l], blo[r]); bhi[i] = np.maximum(bhi[l], bhi[r])
        cnt[i] = cnt[l] + cnt[r]
o = O.OracleScene("scenes/cfg2_smooth_250k.scene", W, H)
scale, aspect, M, cpos = o.camera(); M = M.reshape(4, 4).astype(np.float64)
lights = np.array([[0, 2, -1], [1, -1, -1], [-1, -1, -1]], np.float64)
rng = np.random.default_rng(7)

def primary(tx, ty):
    X, Y = np.meshgrid(tx * 8 + np.arange(8), ty * 8 + np.arange(8))
    xp = (2 * (X.ravel() + 1.0) / W - 1) * scale * aspect; yp = -(2 * (Y.ravel() + 1.0) / H - 1) * scale
    s = np.stack([xp, yp, -np.ones_like(xp)], 1); s /= np.linalg.norm(s, axis=1)[:, None]
    return np.tile(cpos.astype(np.float64), (64, 1)), s @ M[:3, :3] + M[3, :3]

def cert(i, o_, d_):
    U = np.maximum(d_ * mlo[i], d_ * mhi[i]).sum(1); L = np.minimum(d_ * mlo[i], d_ * mhi[i]).sum(1)
    lo = blo[i] - o_; hi = bhi[i] - o_
    behind = (L > 0) & (np.maximum(lo * d_, hi * d_).sum(1) < 0)
    with np.errstate(divide="ignore", invalid="ignore"):
        a = lo / d_; b = hi / d_
    miss = (L > 0) & (np.nanmax(np.minimum(a, b), 1) > np.nanmin(np.maximum(a, b), 1))
    return (U < 0) | behind | miss

def walk(o_, d_, act, smin, smax):
    """returns (node visits, leaf visits, inner certs evaluated) for the wave"""
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / d_
    resume = np.where(act, 0, 1 << 30)
    i = 0; nv = 0; lv = 0; ce = 0
    while i < nN:
        a = resume <= i
        if not a.any():
            i = int(resume.min()); continue
        nv += 1
        t0 = (bounds[i, :3] - o_) * inv; t1 = (bounds[i, 3:] - o_) * inv
        ok = np.nanmax(np.minimum(t0, t1), 1) <= np.nanmin(np.maximum(t0, t1), 1)
        p = a & ok
        nxt = skip[i]
        resume = np.where(a & ~ok, nxt, resume)
        if not p.any():
            i = nxt; continue
        if lc[i] >= 0:
            lv += 1; i += 1; continue
        if smin <= cnt[i] <= smax:
            ce += 1
            c = cert(i, o_, d_)
            resume = np.where(p & c, nxt, resume)
            if not (p & ~c).any():
                i = nxt; continue
        i += 1
    return nv, lv, ce

cands = [(tx, ty) for ty in range(512) for tx in range(512) if (tx * 8 - 2048) ** 2 + (ty * 8 - 2048) ** 2 < 1250 ** 2]
configs = [(1 << 40, 0), (16, 64), (64, 256), (256, 1024), (1024, 4096), (64, 4096), (16, 1 << 40)]
tot = {c: np.zeros(3) for c in configs}; waves = 0; t0 = time.time()
for ti in rng.choice(len(cands), NT, replace=False):
    tx, ty = cands[ti]
    po, pd = primary(tx, ty)
    hits, _ = o.probe(np.concatenate([po, pd], 1).astype(np.float32))
    hit = hits[:, 0] > 0; P = po + pd * hits[:, 3].astype(np.float64)[:, None]
    traces = [(po, pd, np.ones(64, bool))]
    for Lp in lights:
        dirs = Lp - P; dirs /= np.linalg.norm(dirs, axis=1)[:, None]
        traces.append((P + dirs * 1e-4, dirs, hit))
    for o_, d_, act in traces:
        if not act.any(): continue
        waves += 1
        for c in configs: tot[c] += walk(o_, d_, act, c[0], c[1])
print("tiles", NT, "waves", waves, "time %.0fs" % (time.time() - t0))
for c in configs:
    print("inner certs for subtrees of %s refs: node visits/wave %.1f, leaf visits/wave %.1f, inner certs/wave %.1f" % (("%d..%d" % c) if c[1] else "none", tot[c][0] / waves, tot[c][1] / waves, tot[c][2] / waves))
