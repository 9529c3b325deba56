"""CPU research tool (not part of the product): estimates, for sampled 8x8-pixel waves of the headline frame, how many
certificate headers and triangle pairs a wave evaluates inside the leaves it visits under different leaf layouts
(reference order vs spatially sorted references, flat chunks vs a range tree).  Exactness of a re-ordered leaf
comes from breaking t ties by the reference's order."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import rendering_amd as RA
from oracle import oracle as O

W = H = 4096
NT = int(sys.argv[1]) if len(sys.argv) > 1 else 150
RA.set_ac_build("host")
g = RA.Scene("scenes/cfg2_smooth_250k.scene", 64, 64)
d = g.bvh(1)
lc = d["leaf_count"]; lb = d["leaf_begin"]; refs = d["refs"]; bounds = d["bounds"].astype(np.float64)
tr = d["tris"][:, :9].reshape(-1, 3, 3).astype(np.float64)
e1 = tr[:, 1] - tr[:, 0]; e2 = tr[:, 2] - tr[:, 0]
mvec = np.cross(e2, e1)
tlo = tr.min(1); thi = tr.max(1); cen = tr.mean(1)
leaves = np.where(lc > 0)[0]
LB = bounds[leaves]
o = O.OracleScene("scenes/cfg2_smooth_250k.scene", W, H)
scale, aspect, M, cpos = o.camera()
M = M.reshape(4, 4).astype(np.float64)
lights = np.array([[0, 2, -1], [1, -1, -1], [-1, -1, -1]], np.float64)
rng = np.random.default_rng(7)


def primary(tx, ty):
    xs = tx * 8 + np.arange(8); ys = ty * 8 + np.arange(8)
    X, Y = np.meshgrid(xs, ys)
    xp = (2 * (X.ravel() + 1.0) / W - 1) * scale * aspect
    yp = -(2 * (Y.ravel() + 1.0) / H - 1) * scale
    s = np.stack([xp, yp, -np.ones_like(xp)], 1); s /= np.linalg.norm(s, axis=1)[:, None]
    dirs = s @ M[:3, :3] + M[3, :3]
    return np.tile(cpos.astype(np.float64), (64, 1)), dirs


def slab_pass(o_, d_, B):
    """reference slab test (no t range) of rays (n,3) vs boxes (m,6) -> (n,m) bool"""
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / d_
        t0 = (B[None, :, :3] - o_[:, None, :]) * inv[:, None, :]
        t1 = (B[None, :, 3:] - o_[:, None, :]) * inv[:, None, :]
    tn = np.minimum(t0, t1); tf = np.maximum(t0, t1)
    return np.nanmax(tn, 2) <= np.nanmin(tf, 2)


class Range:
    __slots__ = ("mlo", "mhi", "blo", "bhi", "n")


def make_range(ids):
    r = Range(); mm = mvec[ids]
    r.mlo = mm.min(0); r.mhi = mm.max(0); r.blo = tlo[ids].min(0); r.bhi = thi[ids].max(0); r.n = len(ids)
    return r


def cert(r, o_, d_):
    """per-lane: True when the range is certainly rejected (back-facing, or facing and behind / missed)"""
    U = np.maximum(d_ * r.mlo, d_ * r.mhi).sum(1); L = np.minimum(d_ * r.mlo, d_ * r.mhi).sum(1)
    back = U < 0
    facing = L > 0
    lo = r.blo - o_; hi = r.bhi - o_
    boxdot = np.maximum(lo * d_, hi * d_).sum(1)
    behind = facing & (boxdot < 0)
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / d_
        a = lo * inv; b = hi * inv
    miss = facing & (np.nanmax(np.minimum(a, b), 1) > np.nanmin(np.maximum(a, b), 1))
    return back | behind | miss


def morton_order(ids):
    c = cen[ids]; ext = np.maximum(c.max(0) - c.min(0), 1e-12)
    q = np.minimum(((c - c.min(0)) / ext * 1023).astype(np.int64), 1023)
    # drop the thinnest axis (a leaf is a surface patch): interleave the two widest
    ax = np.argsort(ext)[1:]
    def spread(v):
        v = (v | (v << 8)) & 0x00FF00FF; v = (v | (v << 4)) & 0x0F0F0F0F; v = (v | (v << 2)) & 0x33333333; v = (v | (v << 1)) & 0x55555555
        return v
    code = spread(q[:, ax[0]]) | (spread(q[:, ax[1]]) << 1)
    return ids[np.argsort(code, kind="stable")]


_cache = {}


def layout(leaf, scheme):
    key = (leaf, scheme)
    if key in _cache: return _cache[key]
    ids = refs[lb[leaf]:lb[leaf] + lc[leaf]]
    if scheme != "ref": ids = morton_order(ids)
    chunks = [ids[k:k + 8] for k in range(0, len(ids), 8)]
    node = dict(root=make_range(ids), chunks=[make_range(c) for c in chunks])
    if scheme == "tree":
        node["groups"] = [(make_range(ids[k:k + 64]), k // 8, min((k + 64 + 7) // 8, len(chunks))) for k in range(0, len(ids), 64)]
    _cache[key] = node
    return node


def leaf_cost(leaf, scheme, o_, d_, lanes):
    """(headers, triangle pairs) a wave spends in this leaf; lanes = bool mask of lanes that passed the leaf box"""
    L = layout(leaf, scheme)
    n = L["root"].n
    Hn = 1
    live = lanes & ~cert(L["root"], o_, d_)
    if not live.any(): return Hn, 0
    if n <= 8: return Hn, (n + 1) // 2
    T = 0
    if scheme == "tree" and n > 64:
        for gr, c0, c1 in L["groups"]:
            Hn += 1
            lg = live & ~cert(gr, o_, d_)
            if not lg.any(): continue
            for c in L["chunks"][c0:c1]:
                Hn += 1
                if (lg & ~cert(c, o_, d_)).any(): T += (c.n + 1) // 2
        return Hn, T
    for c in L["chunks"]:
        Hn += 1
        if (live & ~cert(c, o_, d_)).any(): T += (c.n + 1) // 2
    return Hn, T


schemes = ["ref", "morton", "tree"]
tot = {s: np.zeros(2) for s in schemes}
totbig = {s: np.zeros(2) for s in schemes}
nodes_vis = 0; waves = 0
t0 = time.time()
# tiles on the sphere's image: centre (2048, ~2048), radius ~1200 px
cands = [(tx, ty) for ty in range(512) for tx in range(512) if (tx * 8 - 2048) ** 2 + (ty * 8 - 2048) ** 2 < 1250 ** 2]
for ti in rng.choice(len(cands), NT, replace=False):
    tx, ty = cands[ti]
    po, pd = primary(tx, ty)
    rays = np.concatenate([po, pd], 1).astype(np.float32)
    hits, _ = o.probe(rays)
    traces = [(po, pd, np.ones(64, bool))]
    hit = hits[:, 0] > 0 if hits.shape[1] >= 8 else np.zeros(64, bool)
    tval = hits[:, 3].astype(np.float64)
    P = po + pd * tval[:, None]
    for Lp in lights:
        dirs = Lp - P; dirs /= np.linalg.norm(dirs, axis=1)[:, None]
        traces.append((P + dirs * 1e-4, dirs, hit))
    for o_, d_, act in traces:
        if not act.any(): continue
        waves += 1
        ps = slab_pass(o_, d_, LB) & act[:, None]
        vis = np.where(ps.any(0))[0]
        for li in vis:
            leaf = leaves[li]
            for s in schemes:
                h, t = leaf_cost(leaf, s, o_, d_, ps[:, li])
                tot[s] += (h, t)
                if lc[leaf] >= 128: totbig[s] += (h, t)
print("tiles", NT, "waves", waves, "time %.0fs" % (time.time() - t0))
for s in schemes:
    print("%-7s headers/wave %.1f  pairs/wave %.1f   | leaves >=128 refs: headers %.1f pairs %.1f" % (s, tot[s][0] / waves, tot[s][1] / waves, totbig[s][0] / waves, totbig[s][1] / waves))
