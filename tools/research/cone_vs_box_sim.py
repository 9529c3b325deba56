"""CPU research tool: wave-level chunk skipping with the box bound on m = v0v2 x v0v1 (what the kernel uses) vs a cone bound
(axis a, |n_k - a| <= delta, |m_k| in [mmin, mmax]) vs both, on Morton-ordered chunks of 8, for sampled waves."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import rendering_amd as RA
from oracle import oracle as O
import importlib.util
spec = importlib.util.spec_from_file_location("los", os.path.join(os.path.dirname(__file__), "leaf_order_sim.py"))
W = H = 4096
NT = int(sys.argv[1]) if len(sys.argv) > 1 else 40
RA.set_ac_build("host")
g = RA.Scene("scenes/cfg2_smooth_250k.scene", 64, 64)
d = g.bvh(1)
lc = d["leaf_count"]; lb = d["leaf_begin"]; refs = d["refs"]; bounds = d["bounds"].astype(np.float64)
tr = d["tris"][:, :9].reshape(-1, 3, 3).astype(np.float64)
mvec = np.cross(tr[:, 2] - tr[:, 0], tr[:, 1] - tr[:, 0]); mlen = np.linalg.norm(mvec, axis=1); nrm = mvec / np.maximum(mlen, 1e-30)[:, None]
tlo = tr.min(1); thi = tr.max(1); cen = tr.mean(1)
leaves = np.where(lc > 0)[0]; LB = bounds[leaves]
o = O.OracleScene("scenes/cfg2_smooth_250k.scene", W, H)
scale, aspect, M, cpos = o.camera(); M = M.reshape(4, 4).astype(np.float64)
lights = np.array([[0, 2, -1], [1, -1, -1], [-1, -1, -1]], np.float64)
rng = np.random.default_rng(7)

def primary(tx, ty):
    X, Y = np.meshgrid(tx * 8 + np.arange(8), ty * 8 + np.arange(8))
    xp = (2 * (X.ravel() + 1.0) / W - 1) * scale * aspect; yp = -(2 * (Y.ravel() + 1.0) / H - 1) * scale
    s = np.stack([xp, yp, -np.ones_like(xp)], 1); s /= np.linalg.norm(s, axis=1)[:, None]
    return np.tile(cpos.astype(np.float64), (64, 1)), s @ M[:3, :3] + M[3, :3]

def slab_pass(o_, d_, B):
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / d_
        t0 = (B[None, :, :3] - o_[:, None, :]) * inv[:, None, :]; t1 = (B[None, :, 3:] - o_[:, None, :]) * inv[:, None, :]
    return np.nanmax(np.minimum(t0, t1), 2) <= np.nanmin(np.maximum(t0, t1), 2)

def morton(ids):
    c = cen[ids]; ext = max((c.max(0) - c.min(0)).max(), 1e-12)
    q = np.minimum(((c - c.min(0)) / ext * 1023).astype(np.int64), 1023)
    def sp(v):
        v = (v | (v << 16)) & 0x030000FF; v = (v | (v << 8)) & 0x0300F00F; v = (v | (v << 4)) & 0x030C30C3; return (v | (v << 2)) & 0x09249249
    return ids[np.argsort(sp(q[:, 0]) | sp(q[:, 1]) << 1 | sp(q[:, 2]) << 2, kind="stable")]

cache = {}
def chunks_of(leaf):
    if leaf in cache: return cache[leaf]
    ids = refs[lb[leaf]:lb[leaf] + lc[leaf]]
    if len(ids) > 8: ids = morton(ids)
    out = []
    for k in range(0, len(ids), 8):
        c = ids[k:k + 8]; mm = mvec[c]; nn = nrm[c]
        a = nn.mean(0); a /= max(np.linalg.norm(a), 1e-30)
        out.append(dict(mlo=mm.min(0), mhi=mm.max(0), blo=tlo[c].min(0), bhi=thi[c].max(0), a=a, delta=np.linalg.norm(nn - a, axis=1).max(),
                        mmin=mlen[c].min(), mmax=mlen[c].max(), n=len(c)))
    cache[leaf] = out
    return out

def geo(ch, o_, d_, facing):
    lo = ch["blo"] - o_; hi = ch["bhi"] - o_
    behind = facing & (np.maximum(lo * d_, hi * d_).sum(1) < 0)
    with np.errstate(divide="ignore", invalid="ignore"):
        a = lo / d_; b = hi / d_
    return behind | (facing & (np.nanmax(np.minimum(a, b), 1) > np.nanmin(np.maximum(a, b), 1)))

def skip_masks(ch, o_, d_):
    U = np.maximum(d_ * ch["mlo"], d_ * ch["mhi"]).sum(1); L = np.minimum(d_ * ch["mlo"], d_ * ch["mhi"]).sum(1)
    box = (U < 0) | geo(ch, o_, d_, L > 0)
    da = d_ @ ch["a"]; dl = np.linalg.norm(d_, axis=1) * ch["delta"]
    cone = (da + dl < 0) | geo(ch, o_, d_, da - dl > 0)
    both = ((U < 0) | (da + dl < 0)) | geo(ch, o_, d_, (L > 0) | (da - dl > 0))
    return box, cone, both

tot = np.zeros(3); nch = 0; waves = 0
cands = [(tx, ty) for ty in range(512) for tx in range(512) if (tx * 8 - 2048) ** 2 + (ty * 8 - 2048) ** 2 < 1250 ** 2]
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
        ps = slab_pass(o_, d_, LB) & act[:, None]
        for li in np.where(ps.any(0))[0]:
            lanes = ps[:, li]
            for ch in chunks_of(leaves[li]):
                nch += 1
                for k, m in enumerate(skip_masks(ch, o_, d_)):
                    if (lanes & ~m).any(): tot[k] += (ch["n"] + 1) // 2
print("waves", waves, "chunk evaluations/wave %.1f" % (nch / waves))
for k, nm in enumerate(("box (kernel)", "cone", "box OR cone")):
    print("%-14s pairs tested per wave %.1f" % (nm, tot[k] / waves))
