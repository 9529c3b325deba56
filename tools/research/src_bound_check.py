"""The bound behind the SOURCE copies of the prune records (csrc/rtx_source.hip, sourceP): when a ray passes within
sigma of a point S -- it starts there (the camera) or ends there (a point light) -- a hit the reference ACCEPTS (objects.cpp:59-95,
restated in numpy fp32, no FMA) lies within  216 dmax ainf P_S + 2^-17 (ainf + |orig|)  of its triangle's box, P_S from the very
function the device runs (rtx_source_p_probe).  Adversarial pairs: the source a few thresholds above the triangle's plane, rays
aimed at and around the triangle, slivers, distances 0.05 ... 30; reports the worst observed distance over the bound and how far
the unconditional Pgen would have allowed.
python tools/research/src_bound_check.py [seed]      (tests/test_prune_bound_cpu.py runs a smaller sample)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import rendering_amd as RA

f32 = np.float32
u = 2.0 ** -24


def reference_mt(o, d, v0, e1, e2, cull=True):
    """Triangle::rayTriangleIntersect (objects.cpp:59-95), one ray per triangle, fp32; cull = options::useBackfaceCulling (objects.cpp:75-79)."""
    px = d[:, 1] * e2[:, 2] - d[:, 2] * e2[:, 1]; py = d[:, 2] * e2[:, 0] - d[:, 0] * e2[:, 2]; pz = d[:, 0] * e2[:, 1] - d[:, 1] * e2[:, 0]
    det = e1[:, 0] * px + e1[:, 1] * py + e1[:, 2] * pz
    ok = ~(det.astype(np.float64) < 1e-8) if cull else ~(np.abs(det.astype(np.float64)) < 1e-8)
    with np.errstate(all="ignore"):
        inv = f32(1) / det
        tx = o[:, 0] - v0[:, 0]; ty = o[:, 1] - v0[:, 1]; tz = o[:, 2] - v0[:, 2]
        uu = (tx * px + ty * py + tz * pz) * inv
        ok &= ~((uu < 0) | (uu > 1))
        qx = ty * e1[:, 2] - tz * e1[:, 1]; qy = tz * e1[:, 0] - tx * e1[:, 2]; qz = tx * e1[:, 1] - ty * e1[:, 0]
        vv = (d[:, 0] * qx + d[:, 1] * qy + d[:, 2] * qz) * inv
        ok &= ~((vv < 0) | (uu + vv > 1))
        tt = (e2[:, 0] * qx + e2[:, 1] * qy + e2[:, 2] * qz) * inv
        ok &= ~(tt < 0)
    return ok, tt, det


def run(N, mode, cam, rng, quiet=False, cull=True):
    # triangles
    sc = 10.0 ** rng.uniform(-3, -1, (N, 1))
    v0 = rng.uniform(-1, 1, (N, 3)) * 10.0 ** rng.uniform(-1, 0.5, (N, 1))
    e1 = rng.normal(size=(N, 3)) * sc; e2 = rng.normal(size=(N, 3)) * sc * 10.0 ** rng.uniform(-1.5, 0, (N, 1))
    if mode == "sliver":
        e2 = e1 * rng.uniform(0.2, 1.5, (N, 1)) + rng.normal(size=(N, 3)) * sc * 10.0 ** rng.uniform(-3, -1, (N, 1))
    v0 = v0.astype(f32); e1 = e1.astype(f32); e2 = e2.astype(f32)
    E1 = e1.astype(np.float64); E2 = e2.astype(np.float64); V0 = v0.astype(np.float64)
    m = np.cross(E2, E1); mm = np.linalg.norm(m, axis=1); nn = m / mm[:, None]
    s1 = np.abs(E1).sum(1); s2 = np.abs(E2).sum(1)
    # the source: at distance D from the triangle, at height H over its plane on the FRONT side (det > 0 <=> dir . m > 0 <=> the ray
    # travels along +m: the source of a ray that starts there lies on the -m side; a light the ray ends at lies on the +m side)
    D = 10.0 ** rng.uniform(-1.3, 1.4, (N, 1))
    # heights from well below the certificate's threshold (no certificate: P_S = Pgen) to comfortable
    H = D * 10.0 ** rng.uniform(-5.5, 0, (N, 1)) if mode != "generic" else D * rng.uniform(0.05, 1, (N, 1))
    tang = E1 * rng.normal(size=(N, 1)) + E2 * rng.normal(size=(N, 1)); tang /= np.linalg.norm(tang, axis=1)[:, None]
    cen = V0 + (E1 + E2) / 3
    inpl = np.sqrt(np.maximum(D * D - H * H, 0))
    side = -1.0 if cam else 1.0
    if not cull:      # culling off: the source on either side of the plane (back faces are accepted too)
        side = side * rng.choice([-1.0, 1.0], (N, 1))
    S = cen + tang * inpl + side * nn * H
    sigma = 0.0 if cam else float(10.0 ** rng.uniform(-5, -3.5))
    # target: in or around the triangle, out to a few edge lengths
    a_, b_ = rng.uniform(-0.5, 1.5, (2, N, 1))
    far_ = rng.random((N, 1)) < 0.3
    a_ = np.where(far_, rng.uniform(-4, 5, (N, 1)), a_); b_ = np.where(far_, rng.uniform(-4, 5, (N, 1)), b_)
    tgt = V0 + a_ * E1 + b_ * E2
    if cam:
        o = S.astype(f32)              # the rays start AT the (fp32) source
        S = o.astype(np.float64)
        dd = tgt - S
        d = (dd / np.linalg.norm(dd, axis=1)[:, None]).astype(f32)
    else:
        # a shadow ray: from a point P on the far side of the triangle towards the light; it passes the light within sigma
        Sp = S + rng.normal(size=(N, 3)) * sigma / np.sqrt(3) * rng.random((N, 1))
        dd = Sp - tgt
        dist = np.linalg.norm(dd, axis=1)[:, None]
        dn = dd / dist
        back = 10.0 ** rng.uniform(-3, 1, (N, 1))
        o = (tgt - dn * back).astype(f32)
        dd2 = Sp - o.astype(np.float64)
        d = (dd2 / np.linalg.norm(dd2, axis=1)[:, None]).astype(f32)
        # (the fp32 origin / direction move the line off Sp: part of what sigma has to cover -- keep only rays that really pass S within sigma)
        o6 = o.astype(np.float64); d6 = d.astype(np.float64)
        w = S - o6
        perp = w - d6 * ((w * d6).sum(1) / (d6 * d6).sum(1))[:, None]
        keep = np.linalg.norm(perp, axis=1) <= sigma
        o, d, v0, e1, e2, S, V0, E1, E2, s1, s2, mm = [x[keep] for x in (o, d, v0, e1, e2, S, V0, E1, E2, s1, s2, mm)]
    ok, tt, det = reference_mt(o, d, v0, e1, e2, cull)
    idx = np.nonzero(ok)[0]
    if len(idx) == 0:
        return 0.0, 0, 0.0, 0.0
    o6 = o[idx].astype(np.float64); d6 = d[idx].astype(np.float64); t6 = tt[idx].astype(np.float64)
    X = o6 + t6[:, None] * d6
    A = V0[idx]; B = A + E1[idx]; Cc = A + E2[idx]
    lo = np.minimum(np.minimum(A, B), Cc); hi = np.maximum(np.maximum(A, B), Cc)
    out = np.maximum(np.maximum(lo - X, X - hi), 0).max(1)
    dmax = np.abs(d6).max(1)
    # ainf as pruneAlive has it for a slot holding just this triangle: to the far side of its box
    ainf = np.maximum(np.abs(o6 - lo), np.abs(o6 - hi)).max(1)
    l2 = (d6 * d6).sum(1)
    assert ((l2 >= 0.9802) & (l2 <= 1.002)).all()
    # (every pair has its own source: one probe call per pair)
    PSv = np.empty(len(idx), np.float32)
    for k, i in enumerate(idx):
        PSv[k] = RA.source_p_probe(v0[i:i + 1], e1[i:i + 1], e2[i:i + 1], S[i], sigma, cam)[0]
    pgen = (s1[idx] * s2[idx]).astype(np.float64)
    use = np.where(ainf <= 32.0, PSv.astype(np.float64), pgen)
    rho = 216.0 * dmax * ainf * use + 2.0 ** -17 * (ainf + np.abs(o6).max(1))
    r = out / rho
    k = int(np.argmax(r))
    cert = PSv < pgen * 0.999
    rho_gen = 216.0 * dmax * ainf * pgen
    if not quiet:
        print("%s %s: accepted %d (with certificate %d); max outside / rho = %.3f (outside %.3e, rho %.3e, det %.2e, P_S %.2e, Pgen %.2e); "
              "certified pairs: max outside / rho %.3f, max outside %.3e, median rho %.2e vs generic %.2e"
              % ("camera" if cam else "light ", mode, len(idx), int(cert.sum()), r.max(), out[k], rho[k], det[idx][k], PSv[k], pgen[k],
                 (out[cert] / rho[cert]).max() if cert.any() else 0, out[cert].max() if cert.any() else 0, np.median(rho[cert]) if cert.any() else 0, np.median(rho_gen[cert]) if cert.any() else 0))
    return float(r.max()), int(cert.sum()), float(out[cert].max() if cert.any() else 0.0), float((out[cert] / rho[cert]).max() if cert.any() else 0.0)


if __name__ == "__main__":
    rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 400000
    for cam in (True, False):
        for mode in ("graze", "sliver", "generic"):
            run(n, mode, cam, rng)
