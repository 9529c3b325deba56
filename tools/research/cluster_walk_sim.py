"""CPU prototype (round 3) of the CLUSTER walk: instead of walking the reference's tree per ray, a wave walks a 64-ary tree over
clusters of 64 Morton-sorted triangles with its whole ray BUNDLE (lane = child record), and only the survivors' triangles
meet the per-triangle bundle filter + exact tests.  A cluster can be skipped when
  (reach) no ray of the bundle passes the box around the reference leaves that hold its triangles (objects.cpp:587-631: a
          triangle is only ever tested through a reached leaf),
  (a)     every triangle is certainly back-facing,  (b) certainly behind the origin,  (c) certainly beyond the limit
          -- the plane tests of the filter's first stage, aggregated over the cluster (normal box + plane-offset interval).
Counts what that leaves per trace on the 250k mesh against today's reached leaves / references.
python tools/research/cluster_walk_sim.py [n_tiles] [size]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import rendering_amd as RA
from rendering_amd import assets
from tools.research.bundle_filter_sim import slab_pass, mt_exact, bundle_filter

f32 = np.float32
CL = int(os.environ.get("CL", "64"))
FAN = int(os.environ.get("FAN", "64"))


def morton3(c, lo, hi):
    q = np.clip(((c - lo) / (hi - lo) * 1023.0), 0, 1023).astype(np.uint64)
    def spread(v):
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        v = (v | (v << 2)) & 0x09249249
        return v
    return spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)


def main():
    n_tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    assets.ensure(["bumpy_250k.obj"])
    RA.set_ac_build("host")
    g = RA.Scene("scenes/cfg2_smooth_250k.scene", S, S)
    b = g.bvh(1)
    tris = b["tris"].astype(np.float64); A = tris[:, 0:3]; B = tris[:, 3:6]; C = tris[:, 6:9]
    E1 = B - A; E2 = C - A
    nT = len(A)
    leaves = np.nonzero(b["leaf_count"] >= 0)[0]
    llo = b["bounds"][leaves, 0:3]; lhi = b["bounds"][leaves, 3:6]
    lbeg = b["leaf_begin"][leaves]; lcnt = b["leaf_count"][leaves]
    refs = b["refs"]
    # per triangle: box around the own boxes of the leaves that hold it
    rlo = np.full((nT, 3), np.inf); rhi = np.full((nT, 3), -np.inf)
    leaf_of_ref = np.repeat(np.arange(len(leaves)), lcnt)
    order = np.argsort(lbeg, kind="stable")
    assert (lbeg[order][1:] >= lbeg[order][:-1]).all()
    leaf_of_ref = np.repeat(order, lcnt[order])
    np.minimum.at(rlo, refs, llo[leaf_of_ref]); np.maximum.at(rhi, refs, lhi[leaf_of_ref])
    print("refs per triangle: mean %.2f max %d" % (len(refs) / nT, np.bincount(refs).max()))
    # clusters
    cen = (A + B + C) / 3
    mo = morton3(cen, cen.min(0), cen.max(0) + 1e-9)
    perm = np.argsort(mo, kind="stable")
    nC = (nT + CL - 1) // CL
    m = np.cross(E2, E1)
    s1 = np.abs(E1).sum(1); s2 = np.abs(E2).sum(1)
    q = m / (s1 * s2)[:, None]
    w = (A * q).sum(1)
    def agg(ids_list):
        out = []
        for ids in ids_list:
            out.append((rlo[ids].min(0), rhi[ids].max(0), q[ids].min(0), q[ids].max(0), w[ids].min(), w[ids].max(),
                        np.minimum(np.minimum(A[ids], B[ids]), C[ids]).min(0), np.maximum(np.maximum(A[ids], B[ids]), C[ids]).max(0)))
        return [np.array(x) for x in zip(*out)]
    cl_ids = [perm[i * CL:(i + 1) * CL] for i in range(nC)]
    c_rlo, c_rhi, c_qlo, c_qhi, c_wlo, c_whi, c_tlo, c_thi = agg(cl_ids)
    nN = (nC + FAN - 1) // FAN
    nd_ids = [np.concatenate(cl_ids[i * FAN:(i + 1) * FAN]) for i in range(nN)]
    n_rlo, n_rhi, n_qlo, n_qhi, n_wlo, n_whi, n_tlo, n_thi = agg(nd_ids)
    print("%d triangles, %d clusters of %d, %d nodes of %d clusters; reach-box size (mean) %s true-box size %s" %
          (nT, nC, CL, nN, FAN, (c_rhi - c_rlo).mean(0), (c_thi - c_tlo).mean(0)))

    scale, aspect, M, pos = g.camera()
    M = M.reshape(4, 4)
    rng = np.random.default_rng(1)
    lights = np.array([[0, 2, -1], [1, -1, -1], [-1, -1, -1]], f32)
    Kf = 2.0 ** -18

    def bundle_line_box(o, d, lo, hi):
        """conservative: may some ray (o in obox, d in dbox) pass the box (line test)?  [n] bool"""
        olo, ohi = o.min(0).astype(np.float64), o.max(0).astype(np.float64)
        with np.errstate(all="ignore"):
            inv = 1.0 / d.astype(np.float64)
        ilo, ihi = inv.min(0), inv.max(0)
        ent = np.full(len(lo), -np.inf); ext = np.full(len(lo), np.inf)
        for k in range(3):
            if not (ilo[k] > 0 or ihi[k] < 0) or not np.isfinite(ilo[k]) or not np.isfinite(ihi[k]):
                # direction sign not uniform: only the origin range constrains (if d_k == 0 the ray must start inside the slab) -- skip
                continue
            if ilo[k] > 0:
                e0 = lo[:, k] - ohi[k]; x0 = hi[:, k] - olo[k]
            else:
                e0 = hi[:, k] - olo[k]; x0 = lo[:, k] - ohi[k]
            eL = np.minimum(e0 * ilo[k], e0 * ihi[k]); xU = np.maximum(x0 * ilo[k], x0 * ihi[k])
            ent = np.maximum(ent, eL); ext = np.minimum(ext, xU)
        return ent <= ext

    def plane_tests(o, d, tmax, qlo, qhi, wlo, whi):
        """(a) back-facing, (b) behind, (c) beyond: [n] bool dead"""
        olo, ohi = o.min(0).astype(np.float64), o.max(0).astype(np.float64)
        dlo, dhi = d.min(0).astype(np.float64), d.max(0).astype(np.float64)
        dmax = max(np.abs(dlo).max(), np.abs(dhi).max())
        def ival_dot(xlo, xhi, qlo, qhi):
            c = np.stack([xlo[None] * qlo, xlo[None] * qhi, xhi[None] * qlo, xhi[None] * qhi])
            return c.min(0).sum(1), c.max(0).sum(1)
        dq_lo, dq_hi = ival_dot(dlo, dhi, qlo, qhi)
        oq_lo, oq_hi = ival_dot(olo, ohi, qlo, qhi)
        ainf = np.maximum(np.abs(olo), np.abs(ohi)).max() + 4.0      # crude
        back = dq_hi + Kf * dmax < 0                                   # det + Ed < 0 for all
        # Nt' = -(a.q) = w - o.q ; behind: Nt' + Et' < 0
        nt_hi = whi - oq_lo
        behind = nt_hi + Kf * ainf < 0
        # beyond: Nt' - Et' >= tmax (det' + Ed')(1+eps)
        tm = float(tmax.max())
        nt_lo = wlo - oq_hi
        beyond = (nt_lo - Kf * ainf >= tm * (dq_hi + Kf * dmax) * (1 + 2.0 ** -18)) & (dq_hi + Kf * dmax > 0) if np.isfinite(tm) and tm < 1e30 else np.zeros(len(qlo), bool)
        return back, behind, beyond

    tot = dict(traces=0, leaves=0, refs=0, passes=0, nodes_alive=0, cl_reach=0, cl_alive=0, cl_alive_true=0, tri_alive=0, missed=0, accepted=0,
               back=0, behind=0, beyond=0)

    def trace(o, d, tmax):
        reach = slab_pass(o, d, llo, lhi)
        L = np.nonzero(reach.any(0) & (lcnt > 0))[0]
        nrefs = int(lcnt[L].sum())
        tot["traces"] += 1; tot["leaves"] += len(L); tot["refs"] += nrefs; tot["passes"] += (nrefs + 63) // 64
        # reference result
        best_t = np.full(len(o), np.inf, f32); best_tri = np.full(len(o), -1)
        acc_tris = set()
        for li in L:
            r = refs[lbeg[li]:lbeg[li] + lcnt[li]]
            ok, t = mt_exact(o, d, A[r].astype(f32), E1[r].astype(f32), E2[r].astype(f32))
            ok &= reach[:, li][:, None]; ok &= t < tmax[:, None]
            for tr in r[ok.any(0)]:
                acc_tris.add(int(tr))
            tt = np.where(ok, t, np.inf)
            k = tt.argmin(1); tk = tt[np.arange(len(o)), k]
            upd = tk < best_t
            best_t = np.where(upd, tk, best_t); best_tri = np.where(upd, r[k], best_tri)
        # cluster walk
        nalive = bundle_line_box(o, d, n_rlo, n_rhi)
        ba, be, by = plane_tests(o, d, tmax, n_qlo, n_qhi, n_wlo, n_whi)
        nalive &= ~(ba | be | by)
        tot["nodes_alive"] += int(nalive.sum())
        alive_tris = set()
        for ni in np.nonzero(nalive)[0]:
            cs = np.arange(ni * FAN, min((ni + 1) * FAN, nC))
            al = bundle_line_box(o, d, c_rlo[cs], c_rhi[cs])
            tot["cl_reach"] += int(al.sum())
            ba, be, by = plane_tests(o, d, tmax, c_qlo[cs], c_qhi[cs], c_wlo[cs], c_whi[cs])
            tot["back"] += int((al & ba).sum()); tot["behind"] += int((al & ~ba & be).sum()); tot["beyond"] += int((al & ~ba & ~be & by).sum())
            al2 = al & ~(ba | be | by)
            tot["cl_alive"] += int(al2.sum())
            al3 = al2 & bundle_line_box(o, d, c_tlo[cs], c_thi[cs])
            tot["cl_alive_true"] += int(al3.sum())
            for c in cs[al2]:
                ids = cl_ids[c]
                alive = bundle_filter(o, d, tmax, A[ids].astype(f32), E1[ids].astype(f32), E2[ids].astype(f32))
                tot["tri_alive"] += int(alive.sum())
                for tr in ids:
                    alive_tris.add(int(tr))
        miss = acc_tris - alive_tris
        tot["missed"] += len(miss); tot["accepted"] += len(acc_tris)
        return best_t, best_tri

    cx = cy = S / 2
    for it in range(n_tiles):
        rad = 1255 * S / 4096 * (np.sqrt(rng.random()) if it % 3 else 0.97 + 0.04 * rng.random())
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
        before = dict(tot)
        bt, btri = trace(o, d, np.full(64, np.finfo(f32).max, f32))
        hit = btri >= 0
        if hit.sum() >= 8:
            P = o[hit] + d[hit] * bt[hit][:, None]
            n = np.cross(E1[btri[hit]], E2[btri[hit]]); n = (n / np.linalg.norm(n, axis=1)[:, None]).astype(f32)
            for Lp in lights:
                dl = (Lp[None] - P); dist = np.linalg.norm(dl, axis=1).astype(f32); dl = (dl / dist[:, None]).astype(f32)
                so = (P + n * f32(1e-4)).astype(f32)
                trace(so, dl, dist)
        dlt = {k: tot[k] - before[k] for k in tot}
        print("tile (%d,%d) r=%.0f: traces %d | today leaves %d refs %d passes %d | cluster walk: nodes %d, clusters reach %d (-back %d -behind %d -beyond %d) alive %d (with true box %d), filter survivors %d, missed %d" %
              (tx, ty, rad, dlt["traces"], dlt["leaves"], dlt["refs"], dlt["passes"], dlt["nodes_alive"], dlt["cl_reach"], dlt["back"], dlt["behind"], dlt["beyond"], dlt["cl_alive"], dlt["cl_alive_true"], dlt["tri_alive"], dlt["missed"]))
    n = tot["traces"]
    print({k: round(v / n, 2) for k, v in tot.items()})


if __name__ == "__main__":
    main()
