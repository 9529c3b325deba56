"""CPU prototype of the bundle filter: for an 8x8-pixel tile of rays, every triangle of every
leaf some ray reaches is classified against the BUNDLE (box of origins x box of directions) with division-free tests on
the Moller-Trumbore numerators; only the survivors would be tested per ray.  Checks on the 250k mesh that no triangle
the reference accepts for any ray of the tile is filtered out, and counts the survivors.
python tools/research/bundle_filter_sim.py [n_tiles]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import rendering_amd as RA
from rendering_amd import assets

f32 = np.float32
K = f32(2.0 ** -18)
ETA = f32(1e-30)


def mt_exact(o, d, v0, e1, e2, cull=True):
    """Triangle::rayTriangleIntersect (objects.cpp:59-95) in fp32, rays [R,3] x triangles [T,3] -> accept [R,T], t."""
    o = o[:, None, :]; d = d[:, None, :]
    v0 = v0[None]; e1 = e1[None]; e2 = e2[None]
    px = d[..., 1] * e2[..., 2] - d[..., 2] * e2[..., 1]
    py = d[..., 2] * e2[..., 0] - d[..., 0] * e2[..., 2]
    pz = d[..., 0] * e2[..., 1] - d[..., 1] * e2[..., 0]
    det = e1[..., 0] * px + e1[..., 1] * py + e1[..., 2] * pz
    ok = ~(det.astype(np.float64) < 1e-8) if cull else ~(np.abs(det).astype(np.float64) < 1e-8)
    with np.errstate(all="ignore"):
        inv = f32(1) / det
        tx = o[..., 0] - v0[..., 0]; ty = o[..., 1] - v0[..., 1]; tz = o[..., 2] - v0[..., 2]
        u = (tx * px + ty * py + tz * pz) * inv
        ok &= ~((u < 0) | (u > 1))
        qx = ty * e1[..., 2] - tz * e1[..., 1]
        qy = tz * e1[..., 0] - tx * e1[..., 2]
        qz = tx * e1[..., 1] - ty * e1[..., 0]
        v = (d[..., 0] * qx + d[..., 1] * qy + d[..., 2] * qz) * inv
        ok &= ~((v < 0) | (u + v > 1))
        t = (e2[..., 0] * qx + e2[..., 1] * qy + e2[..., 2] * qz) * inv
        ok &= ~(t < 0)
    return ok, t


def slab_pass(o, d, lo, hi):
    """intersectBox (objects.cpp:534-570) rays [R,3] x boxes [B,3] -> [R,B] (NaN compares false = pass)."""
    with np.errstate(all="ignore"):
        inv = f32(1) / d
        sgn = inv < 0
        o = o[:, None, :]; inv = inv[:, None, :]; sgn = sgn[:, None, :]
        bmin = np.where(sgn, hi[None], lo[None]); bmax = np.where(sgn, lo[None], hi[None])
        tmn = (bmin - o) * inv; tmx = (bmax - o) * inv
        tmin = tmn[..., 0].copy(); tmax = tmx[..., 0].copy()
        fail = (tmin > tmx[..., 1]) | (tmn[..., 1] > tmax)
        tmin = np.where(tmn[..., 1] > tmin, tmn[..., 1], tmin); tmax = np.where(tmx[..., 1] < tmax, tmx[..., 1], tmax)
        fail |= (tmin > tmx[..., 2]) | (tmn[..., 2] > tmax)
    return ~fail


def bundle_filter(o, d, tmax, v0, e1, e2, cull=True):
    """alive [T]: triangles that MAY be accepted by some ray of the bundle."""
    olo, ohi = o.min(0), o.max(0); dlo, dhi = d.min(0), d.max(0)
    oc = (olo + ohi) * f32(0.5); dc = (dlo + dhi) * f32(0.5)
    ro = (ohi - olo) * f32(0.5) * f32(1 + 2.0 ** -20) + f32(2.0 ** -22) * np.maximum(np.abs(olo), np.abs(ohi))
    rd = (dhi - dlo) * f32(0.5) * f32(1 + 2.0 ** -20) + f32(2.0 ** -22) * np.maximum(np.abs(dlo), np.abs(dhi))
    dmax = (np.abs(dc) + rd).max() * f32(1 + 2.0 ** -20)
    ro_sum_dmax = ro.sum() * dmax * f32(1 + 2.0 ** -20)
    tmaxB = f32(tmax.max())
    a = oc[None] - v0
    ab = np.abs
    m = np.stack([e2[:, 1] * e1[:, 2] - e2[:, 2] * e1[:, 1], e2[:, 2] * e1[:, 0] - e2[:, 0] * e1[:, 2], e2[:, 0] * e1[:, 1] - e2[:, 1] * e1[:, 0]], 1)
    s1 = ab(e1).sum(1); s2 = ab(e2).sum(1)
    ainf = (ab(a).max(1) + ro.max()) * f32(1 + 2.0 ** -20)
    detc = (m * dc[None]).sum(1); detr = (ab(m) * rd[None]).sum(1) * f32(1 + 2.0 ** -20)
    Ed = K * dmax * s1 * s2 + ETA
    detHi = detc + detr + Ed
    Et = K * ainf * s1 * s2 + ETA
    ntc = -(a * m).sum(1); ntr = (ab(m) * ro[None]).sum(1) * f32(1 + 2.0 ** -20)
    one = f32(1 + 2.0 ** -18)
    rej = np.zeros(len(v0), bool)
    if cull:
        rej |= detHi < 0                      # det_c < 1e-8 for every ray
    rej |= ntc + ntr < -Et                    # t_c < 0
    rej |= ntc - ntr - Et >= tmaxB * detHi * one   # t_c >= every lane's limit
    wu = np.stack([e2[:, 1] * a[:, 2] - e2[:, 2] * a[:, 1], e2[:, 2] * a[:, 0] - e2[:, 0] * a[:, 2], e2[:, 0] * a[:, 1] - e2[:, 1] * a[:, 0]], 1)
    wv = np.stack([a[:, 1] * e1[:, 2] - a[:, 2] * e1[:, 1], a[:, 2] * e1[:, 0] - a[:, 0] * e1[:, 2], a[:, 0] * e1[:, 1] - a[:, 1] * e1[:, 0]], 1)
    nuc = (wu * dc[None]).sum(1); nur = ((ab(wu) * rd[None]).sum(1) + ro_sum_dmax * s2) * f32(1 + 2.0 ** -20)
    nvc = (wv * dc[None]).sum(1); nvr = ((ab(wv) * rd[None]).sum(1) + ro_sum_dmax * s1) * f32(1 + 2.0 ** -20)
    Eu = K * dmax * ainf * s2 + ETA; Ev = K * dmax * ainf * s1 + ETA
    rej |= nuc + nur < -Eu
    rej |= nvc + nvr < -Ev
    rej |= nuc - nur - Eu > detHi * one
    rej |= (nuc - nur - Eu) + (nvc - nvr - Ev) > detHi * one
    return ~rej


def main():
    n_tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    assets.ensure(["bumpy_250k.obj"])
    RA.set_ac_build("host")
    S = 4096
    g = RA.Scene("scenes/cfg2_smooth_250k.scene", S, S)
    b = g.bvh(1)
    tris = b["tris"]; A = tris[:, 0:3]; B = tris[:, 3:6]; C = tris[:, 6:9]
    E1 = B - A; E2 = C - A
    leaves = np.nonzero(b["leaf_count"] >= 0)[0]
    llo = b["bounds"][leaves, 0:3]; lhi = b["bounds"][leaves, 3:6]
    lbeg = b["leaf_begin"][leaves]; lcnt = b["leaf_count"][leaves]
    scale, aspect, M, pos = g.camera()
    M = M.reshape(4, 4)
    rng = np.random.default_rng(1)
    lights = np.array([[0, 2, -1], [1, -1, -1], [-1, -1, -1]], f32)
    tot = dict(tiles=0, traces=0, tris=0, alive=0, accepted=0, missed=0, leaves=0)

    def trace(o, d, tmax, tag):
        reach = slab_pass(o, d, llo, lhi)                  # reachability approximated by the leaf's own box (nested boxes)
        L = np.nonzero(reach.any(0))[0]
        best_t = np.full(len(o), np.inf, f32); best_tri = np.full(len(o), -1)
        nt = na = 0
        for li in L:
            r = b["refs"][lbeg[li]:lbeg[li] + lcnt[li]]
            if len(r) == 0:
                continue
            alive = bundle_filter(o, d, tmax, A[r], E1[r], E2[r])
            ok, t = mt_exact(o, d, A[r], E1[r], E2[r])
            ok &= reach[:, li][:, None]
            ok &= t < tmax[:, None]
            bad = ok & ~alive[None]
            tot["missed"] += int(bad.sum()); tot["accepted"] += int(ok.sum())
            nt += len(r); na += int(alive.sum())
            tt = np.where(ok, t, np.inf)
            k = tt.argmin(1); tk = tt[np.arange(len(o)), k]
            upd = tk < best_t
            best_t = np.where(upd, tk, best_t); best_tri = np.where(upd, r[k], best_tri)
        tot["traces"] += 1; tot["tris"] += nt; tot["alive"] += na; tot["leaves"] += len(L)
        return best_t, best_tri, nt, na

    cx = cy = S / 2
    for it in range(n_tiles):
        # tiles over the disc of the mesh (radius ~1255 px), some on the silhouette
        rad = 1255 * (np.sqrt(rng.random()) if it % 3 else 0.97 + 0.04 * rng.random())
        ang = rng.random() * 2 * np.pi
        tx = int((cx + rad * np.cos(ang)) // 8); ty = int((cy + rad * np.sin(ang)) // 8)
        xs, ys = np.meshgrid(np.arange(8) + tx * 8, np.arange(8) + ty * 8)
        x = xs.ravel().astype(f32) + f32(1.0); y = ys.ravel().astype(f32) + f32(1.0)
        xp = (f32(2) * x / f32(S) - f32(1)) * scale * aspect
        yp = -(f32(2) * y / f32(S) - f32(1)) * scale
        s = np.stack([xp, yp, -np.ones_like(xp)], 1)
        s = (s * (f32(1) / np.sqrt((s.astype(np.float64) ** 2).sum(1))).astype(f32)[:, None]).astype(f32)
        d = (s @ M[:3, :3] + M[3, :3]).astype(f32)
        o = np.repeat(pos[None].astype(f32), 64, 0)
        bt, btri, nt, na = trace(o, d, np.full(64, np.finfo(f32).max, f32), "primary")
        line = "tile (%d,%d) r=%.0f primary: %d tris in reached leaves, %d survive the bundle filter, %d lanes hit" % (tx, ty, rad, nt, na, int((btri >= 0).sum()))
        hit = btri >= 0
        if hit.sum() >= 8:
            P = o[hit] + d[hit] * bt[hit][:, None]
            n = np.cross(E1[btri[hit]], E2[btri[hit]]); n = (n / np.linalg.norm(n, axis=1)[:, None]).astype(f32)
            for Lp in lights:
                dl = (Lp[None] - P); dist = np.linalg.norm(dl, axis=1).astype(f32); dl = (dl / dist[:, None]).astype(f32)
                so = (P + n * f32(1e-4)).astype(f32)
                _, stri, nt, na = trace(so, dl, dist, "shadow")
                line += " | shadow %d/%d occl %d" % (na, nt, int((stri >= 0).sum()))
        print(line)
        tot["tiles"] += 1
    print(tot)
    print("survivors per trace %.1f of %.0f triangles in reached leaves (%.2f%%); accepted (ray, triangle) pairs %d, filtered-out accepted pairs %d"
          % (tot["alive"] / tot["traces"], tot["tris"] / tot["traces"], 100.0 * tot["alive"] / max(tot["tris"], 1), tot["accepted"], tot["missed"]))


if __name__ == "__main__":
    main()
