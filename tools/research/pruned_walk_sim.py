"""CPU prototype (round 3): what an ORDERED, PRUNED walk of the reference's own tree would save per Render::trace of a wave.
The wave walks the binary tree (a node is visited when some open ray passes its box; boxes are nested, so that is the
reference's reachability), children nearest-first along the bundle's direction, and a subtree is skipped when
  plane : every triangle in it is certainly back-facing / behind the origin / beyond every ray's limit -- the first-stage
          tests of the bundle filter aggregated over the subtree (box of scaled normals q = m / (s1 s2), interval of plane offsets
          w = v0 . q): rigorous whatever the conditioning;
  tbox  : every open ray's segment [0, limit] misses the TRUE box of the subtree's triangles (needs a margin
          argument for badly conditioned pairs).
python tools/research/pruned_walk_sim.py [n_tiles] [size]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import rendering_amd as RA
from rendering_amd import assets
from tools.research.bundle_filter_sim import slab_pass, mt_exact, bundle_filter

f32 = np.float32
Kf = 2.0 ** -18


def main():
    n_tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    scene = os.environ.get("SCENE", "scenes/cfg2_smooth_250k.scene")
    assets.ensure(["bumpy_250k.obj"])
    RA.set_ac_build("host")
    g = RA.Scene(scene, S, S)
    b = g.bvh(int(os.environ.get("OBJ", "1")))
    tris = b["tris"].astype(np.float64); A = tris[:, 0:3]; B = tris[:, 3:6]; C = tris[:, 6:9]
    E1 = B - A; E2 = C - A
    nN = b["n_nodes"]
    lo = b["bounds"][:, 0:3].astype(np.float64); hi = b["bounds"][:, 3:6].astype(np.float64)
    skip = b["skip"]; lbeg = b["leaf_begin"]; lcnt = b["leaf_count"]; refs = b["refs"]
    is_leaf = lcnt >= 0
    m = np.cross(E2, E1); s1 = np.abs(E1).sum(1); s2 = np.abs(E2).sum(1)
    q = m / (s1 * s2)[:, None]; w = (A * q).sum(1)
    tlo_t = np.minimum(np.minimum(A, B), C); thi_t = np.maximum(np.maximum(A, B), C)
    inf = np.inf
    qlo = np.full((nN, 3), inf); qhi = np.full((nN, 3), -inf); wlo = np.full(nN, inf); whi = np.full(nN, -inf)
    tlo = np.full((nN, 3), inf); thi = np.full((nN, 3), -inf)
    for i in range(nN - 1, -1, -1):
        if is_leaf[i]:
            r = refs[lbeg[i]:lbeg[i] + lcnt[i]]
            if len(r):
                qlo[i] = q[r].min(0); qhi[i] = q[r].max(0); wlo[i] = w[r].min(); whi[i] = w[r].max()
                tlo[i] = tlo_t[r].min(0); thi[i] = thi_t[r].max(0)
        else:
            l, r_ = i + 1, skip[i + 1]
            qlo[i] = np.minimum(qlo[l], qlo[r_]); qhi[i] = np.maximum(qhi[l], qhi[r_])
            wlo[i] = min(wlo[l], wlo[r_]); whi[i] = max(whi[l], whi[r_])
            tlo[i] = np.minimum(tlo[l], tlo[r_]); thi[i] = np.maximum(thi[l], thi[r_])
    depth = np.zeros(nN, int)
    for i in range(nN):
        if not is_leaf[i]:
            depth[i + 1] = depth[i] + 1; depth[skip[i + 1]] = depth[i] + 1
    hist_visit = np.zeros(40, int); hist_prune = np.zeros(40, int)
    chunk_stat = [0, 0, 0, 0]      # chunks of 64 references in reached leaves, of which a chunk record would prune; the same for leaves of more than one chunk
    # round 4: nodes reached (some open ray passes their box) by binary depth, per mode -- the number of WIDE-node visits if g binary levels share a fetch is about the
    # sum over the depths that are multiples of g (the kernel walks g = 2 today; DESIGN.md section 7.1 asks what g = 3 / 4 would save)
    reached_by_depth = {}
    scale, aspect, M, pos = g.camera()
    M = M.reshape(4, 4)
    rng = np.random.default_rng(1)
    lights = np.array([[0, 2, -1], [1, -1, -1], [-1, -1, -1]], f32)

    def plane_dead(o, d, tm, i):
        if not np.isfinite(wlo[i]):
            return True          # no triangles at all
        olo, ohi = o.min(0).astype(np.float64), o.max(0).astype(np.float64)
        dlo, dhi = d.min(0).astype(np.float64), d.max(0).astype(np.float64)
        dmax = max(np.abs(dlo).max(), np.abs(dhi).max())
        def ival(xlo, xhi):
            c = np.stack([xlo * qlo[i], xlo * qhi[i], xhi * qlo[i], xhi * qhi[i]])
            return c.min(0).sum(), c.max(0).sum()
        dq_lo, dq_hi = ival(dlo, dhi); oq_lo, oq_hi = ival(olo, ohi)
        ainf = np.maximum(np.abs(olo - tlo[i]), np.abs(ohi - thi[i])).max() * 1.01
        if dq_hi + Kf * dmax < 0:
            return True
        if (whi[i] - oq_lo) + Kf * ainf < 0:
            return True
        if tm < 1e30 and (wlo[i] - oq_hi) - Kf * ainf >= tm * (dq_hi + Kf * dmax) * (1 + 2.0 ** -18):
            return True
        return False

    def seg_box(o, inv, lim, blo, bhi, margin):
        """per ray: does the segment [0, lim] pass the box (inflated)?"""
        with np.errstate(all="ignore"):
            t0 = (blo[None] - margin - o) * inv; t1 = (bhi[None] + margin - o) * inv
            ent = np.minimum(t0, t1).max(1); ext = np.maximum(t0, t1).min(1)
        return (ent <= ext) & (ext >= 0) & (ent <= lim)

    def cert_rho(o, d, i):
        """per ray: rho = 36 u dmax ainf / (dq_lo - 6 u dmax) where dq_lo = min over the subtree's normal box of d . q (inf: no certificate)"""
        d64 = d.astype(np.float64)
        c = np.minimum(d64 * qlo[i][None], d64 * qhi[i][None]).sum(1)
        dmax = np.abs(d64).max(1)
        ainf = np.maximum(np.abs(o.astype(np.float64) - tlo[i][None]), np.abs(o.astype(np.float64) - thi[i][None])).max(1)
        u = 2.0 ** -24
        with np.errstate(all="ignore"):
            rho = np.where(c > KAPPA * dmax, 36 * u * dmax * ainf / (c - 6 * u * dmax), np.inf)
        return rho

    def cert_rho_bundle(o, d, i):
        d64 = d.astype(np.float64); o6 = o.astype(np.float64)
        dlo, dhi = d64.min(0), d64.max(0)
        c = np.minimum(np.minimum(dlo * qlo[i], dlo * qhi[i]), np.minimum(dhi * qlo[i], dhi * qhi[i])).sum()
        dmax = np.abs(d64).max()
        ainf = np.maximum(np.abs(o6.min(0) - thi[i]), np.abs(o6.max(0) - tlo[i])).max()
        u = 2.0 ** -24
        r = 36 * u * dmax * ainf / (c - 6 * u * dmax) if c > 2.0 ** -16 * dmax else np.inf
        return np.full(len(o), r)

    def seg_box_rho(o, inv, lim, blo, bhi, rho):
        with np.errstate(all="ignore"):
            t0 = (blo[None] - rho[:, None] - o) * inv; t1 = (bhi[None] + rho[:, None] - o) * inv
            ent = np.minimum(t0, t1).max(1); ext = np.maximum(t0, t1).min(1)
        return ~np.isfinite(rho) | ((ent <= ext) & (ext >= 0) & (ent <= lim * (1 + 2.0 ** -20)))

    KAPPA = float(os.environ.get("KAPPA", str(2.0 ** -12)))
    PS = np.zeros(nN)
    ps_t = s1 * s2
    for i in range(nN - 1, -1, -1):
        if is_leaf[i]:
            r = refs[lbeg[i]:lbeg[i] + lcnt[i]]
            PS[i] = ps_t[r].max() if len(r) else 0.0
        else:
            PS[i] = max(PS[i + 1], PS[skip[i + 1]])
    print("P_S: root %.2e, leaf median %.2e, leaf 90%% %.2e" % (PS[0], np.median(PS[is_leaf & (lcnt > 0)]), np.quantile(PS[is_leaf & (lcnt > 0)], 0.9)))

    def unc_rho(o, d, i):
        d64 = d.astype(np.float64); o6 = o.astype(np.float64)
        dmax = np.abs(d64).max()
        ainf = np.maximum(np.abs(o6.min(0) - thi[i]), np.abs(o6.max(0) - tlo[i])).max()
        return np.full(len(o), 212.0 * 1.05 * dmax * ainf * PS[i] + 2.0 ** -18 * (np.abs(o6).max() + max(np.abs(tlo[i]).max(), np.abs(thi[i]).max())))

    modes = ["today", "plane", "both", "unc", "unc_plane", "mix", "mix_plane"]
    tot = {mo: dict(nodes=0, leaves=0, refs=0, passes=0, wide=0) for mo in modes}
    ntr = [0]
    mism = [0]

    def trace(o, d, tmax, shadow):
        ntr[0] += 1
        with np.errstate(all="ignore"):
            inv = (f32(1) / d).astype(np.float64)
        o64 = o.astype(np.float64)
        dc = d.mean(0)
        results = {}
        for mo in modes:
            best = tmax.astype(np.float64).copy(); btri = np.full(len(o), -1)
            stack = [0]
            nodes = leaves = nrefs = passes = 0
            visited_inner_depth = []
            while stack:
                i = stack.pop()
                open_ = np.ones(len(o), bool) if not shadow else (btri < 0)
                if not open_.any():
                    break
                # reachability (own box)
                p = slab_pass(o[open_], d[open_], lo[i:i + 1].astype(f32), hi[i:i + 1].astype(f32))[:, 0]
                nodes += 1
                if not p.any():
                    continue
                reached_by_depth.setdefault(mo, np.zeros(40, int))[depth[i]] += 1
                if mo == "unc_plane":
                    hist_visit[depth[i]] += 1
                if mo in ("plane", "both", "bothc", "bothc_ord", "unc_plane", "mix_plane") and plane_dead(o[open_], d[open_], best[open_].max(), i):
                    if mo == "unc_plane":
                        hist_prune[depth[i]] += 1
                    continue
                if mo in ("tbox", "both") and np.isfinite(tlo[i]).all():
                    if not seg_box(o64[open_], inv[open_], best[open_], tlo[i], thi[i], float(os.environ.get('MARGIN','1e-3'))).any():
                        continue
                if mo in ("tboxc", "bothc", "bothc_ord") and np.isfinite(tlo[i]).all():
                    rho = cert_rho(o[open_], d[open_], i)
                    if not seg_box_rho(o64[open_], inv[open_], best[open_], tlo[i], thi[i], rho).any():
                        continue
                if mo in ("unc", "unc_plane", "mix", "mix_plane") and np.isfinite(tlo[i]).all():
                    rho = unc_rho(o[open_], d[open_], i)
                    if mo.startswith("mix"):
                        rho = np.minimum(rho, cert_rho_bundle(o[open_], d[open_], i))
                    if not seg_box_rho(o64[open_], inv[open_], np.full(int(open_.sum()), best[open_].max()), tlo[i], thi[i], rho).any():
                        if mo == "unc_plane":
                            hist_prune[depth[i]] += 1
                        continue
                if is_leaf[i]:
                    if lcnt[i] == 0:
                        continue
                    leaves += 1; nrefs += lcnt[i]; passes += (lcnt[i] + 63) // 64
                    r = refs[lbeg[i]:lbeg[i] + lcnt[i]]
                    if mo == "mix_plane":
                        # round 4: would a prune record per 64-reference CHUNK of the leaf skip the pass?  (same two tests as for a slot, the chunk's own aggregates)
                        oo, dd = o[open_], d[open_]
                        o6, d6 = oo.astype(np.float64), dd.astype(np.float64)
                        tm = best[open_].max()
                        for c0 in range(0, len(r), 64):
                            rc = r[c0:c0 + 64]
                            chunk_stat[0] += 1
                            if len(r) > 64:
                                chunk_stat[2] += 1
                            cq_lo, cq_hi = q[rc].min(0), q[rc].max(0); cw_lo, cw_hi = w[rc].min(), w[rc].max()
                            ct_lo, ct_hi = tlo_t[rc].min(0), thi_t[rc].max(0)
                            dlo, dhi = d6.min(0), d6.max(0); olo, ohi = o6.min(0), o6.max(0)
                            dmax = max(np.abs(dlo).max(), np.abs(dhi).max())
                            def ival(xlo, xhi):
                                cc = np.stack([xlo * cq_lo, xlo * cq_hi, xhi * cq_lo, xhi * cq_hi])
                                return cc.min(0).sum(), cc.max(0).sum()
                            dq_lo, dq_hi = ival(dlo, dhi); oq_lo, oq_hi = ival(olo, ohi)
                            ainf = np.maximum(np.abs(olo - ct_lo), np.abs(ohi - ct_hi)).max() * 1.01
                            dead = dq_hi + Kf * dmax < 0 or (cw_hi - oq_lo) + Kf * ainf < 0 or (tm < 1e30 and (cw_lo - oq_hi) - Kf * ainf >= tm * (dq_hi + Kf * dmax) * (1 + 2.0 ** -18))
                            if not dead:
                                ainf2 = np.maximum(np.abs(olo - ct_hi), np.abs(ohi - ct_lo)).max()
                                rho = 212.0 * 1.05 * dmax * ainf2 * float(os.environ.get("CHUNK_PS", "1e-3")) * ps_t[rc].max() / max(ps_t[rc].max(), 1e-30) + 2.0 ** -18 * (np.abs(o6).max() + max(np.abs(ct_lo).max(), np.abs(ct_hi).max()))
                                dead = not seg_box_rho(o6, (1.0 / d6), np.full(len(o6), tm), ct_lo, ct_hi, np.full(len(o6), rho)).any()
                            if dead:
                                chunk_stat[1] += 1
                                if len(r) > 64:
                                    chunk_stat[3] += 1
                    reach = np.zeros(len(o), bool); reach[np.nonzero(open_)[0][p]] = True
                    ok, t = mt_exact(o, d, A[r].astype(f32), E1[r].astype(f32), E2[r].astype(f32))
                    ok &= reach[:, None]
                    tt = np.where(ok, t, np.inf)
                    k = tt.argmin(1); tk = tt[np.arange(len(o)), k]
                    upd = tk < best
                    best = np.where(upd, tk, best); btri = np.where(upd, r[k], btri)
                    continue
                l, r_ = i + 1, skip[i + 1]
                if mo != "bothc_ord":
                    stack.append(r_); stack.append(l)
                else:
                    # nearest first: the child whose box starts nearer along the bundle's mean direction
                    ax = int(np.argmax(np.abs(hi[l] - hi[r_]) + np.abs(lo[l] - lo[r_])))
                    left_first = (dc[ax] > 0) == (lo[l][ax] <= lo[r_][ax])
                    if left_first:
                        stack.append(r_); stack.append(l)
                    else:
                        stack.append(l); stack.append(r_)
            tot[mo]["nodes"] += nodes; tot[mo]["leaves"] += leaves; tot[mo]["refs"] += nrefs; tot[mo]["passes"] += passes
            results[mo] = (best.copy(), btri.copy())
        b0, t0 = results["today"]
        for mo in modes[1:]:
            b1, t1 = results[mo]
            if shadow:
                bad = ((t0 >= 0) != (t1 >= 0)).sum()
            else:
                bad = (b0 != b1).sum()
            mism[0] += int(bad)
        return results["today"]

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
        before = {mo: dict(tot[mo]) for mo in modes}
        bt, btri = trace(o, d, np.full(64, np.finfo(f32).max, f32), False)
        hit = btri >= 0
        if hit.sum() >= 8:
            P = o[hit] + d[hit] * bt[hit][:, None].astype(f32)
            n = np.cross(E1[btri[hit]], E2[btri[hit]]); n = (n / np.linalg.norm(n, axis=1)[:, None]).astype(f32)
            for Lp in lights:
                dl = (Lp[None] - P); dist = np.linalg.norm(dl, axis=1).astype(f32); dl = (dl / dist[:, None]).astype(f32)
                facing = (n * dl).sum(1) > 0
                if facing.sum() == 0:
                    continue
                so = (P + n * f32(1e-4)).astype(f32)
                trace(so[facing], dl[facing], dist[facing], True)
        print("tile (%d,%d) r=%.0f hit %d: " % (tx, ty, rad, hit.sum()) + " | ".join("%s n%d l%d p%d" % (mo, tot[mo]["nodes"] - before[mo]["nodes"], tot[mo]["leaves"] - before[mo]["leaves"], tot[mo]["passes"] - before[mo]["passes"]) for mo in modes))
    print("unc_plane: by binary depth: tested / pruned:", [(int(a), int(b)) for a, b in zip(hist_visit, hist_prune)][:30])
    n = ntr[0]
    print("traces", n, "mismatching lanes vs today:", mism[0])
    print("mix_plane: chunks of 64 references in reached leaves %d, a record per chunk would prune %d (%.0f %%); in leaves of more than one chunk: %d, pruned %d (%.0f %%)  [margin: rho with P = CHUNK_PS = %s]" % (
        chunk_stat[0], chunk_stat[1], 100.0 * chunk_stat[1] / max(chunk_stat[0], 1), chunk_stat[2], chunk_stat[3], 100.0 * chunk_stat[3] / max(chunk_stat[2], 1), os.environ.get("CHUNK_PS", "1e-3")))
    for mo, h in reached_by_depth.items():
        print(mo, "nodes reached per trace at the depths that are multiples of g (~ wide-node visits with g levels per fetch): " + ", ".join("g=%d: %.1f" % (gg, h[::gg].sum() / n) for gg in (1, 2, 3, 4)))
    for mo in modes:
        print(mo, {k: round(v / n, 1) for k, v in tot[mo].items()})


if __name__ == "__main__":
    main()
