"""The prune bounds on REAL data: over rows of the headline frame (250k mesh, 4096^2) the hits the reference
accepts -- the closest hit of every primary ray and of the shadow rays towards the three point lights, traced by the CPU oracle
(bit-identical to the reference) -- against the inflation the prune records allow for their triangle: how far the accepted point lies
outside the triangle's box, relative to  rho = 216 dmax ainf P + 2^-17 (ainf + |orig|)  with P = Pgen = s1 s2 (the unconditional
bound) and with P = P_S of the ray's source (the camera / the light: DESIGN.md 3.1d, rtx_source_p_probe).
python tools/research/rho_real_data.py [row stride, default 64] [size, default 4096]  ->  profiles/r04_rho_real_data.txt"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import rendering_amd as RA
from rendering_amd import assets
from oracle import oracle as O

f32 = np.float32
u24 = 2.0 ** -24


def main(stride=64, S=4096, out=sys.stdout):
    assets.ensure(["bumpy_250k.obj"])
    scene = "scenes/cfg2_smooth_250k.scene"
    o = O.OracleScene(scene, S, S)
    b = o.bvh(1)
    tris = b["tris"]
    A = tris[:, 0:3]; E1 = (tris[:, 3:6] - A).astype(f32); E2 = (tris[:, 6:9] - A).astype(f32)      # the fp32 differences of the exact test
    NA, NB, NC = tris[:, 9:12], tris[:, 12:15], tris[:, 15:18]
    scale, aspect, M, pos = o.camera(); M = M.reshape(4, 4)
    lights = np.array([[0, 2, -1], [1, -1, -1], [-1, -1, -1]], f32)
    bias = f32(1e-4)
    ys = np.arange(stride // 2, S - 1, stride)
    xs = np.arange(S - 1)
    X, Y = np.meshgrid(xs, ys)
    x = X.ravel().astype(f32) + f32(1.0); y = Y.ravel().astype(f32) + f32(1.0)
    xp = (f32(2) * x / f32(S) - f32(1)) * scale * aspect
    yp = -(f32(2) * y / f32(S) - f32(1)) * scale
    s = np.stack([xp, yp, -np.ones_like(xp)], 1)
    s = (s * (f32(1) / np.sqrt((s.astype(np.float64) ** 2).sum(1))).astype(f32)[:, None]).astype(f32)
    d = (s @ M[:3, :3] + M[3, :3]).astype(f32)
    org = np.repeat(pos[None].astype(f32), len(d), 0)

    def check(name, ro, rd, hits, S_src, sigma, cam, limit=None):
        hit = (hits[:, 0] > 0) & (hits[:, 1] == 1)
        if limit is not None:
            hit &= hits[:, 3] < limit
        t = hits[hit, 3].astype(np.float64); tri = hits[hit, 2].astype(np.int64)
        o6 = ro[hit].astype(np.float64); d6 = rd[hit].astype(np.float64)
        Xh = o6 + t[:, None] * d6
        a = A[tri].astype(np.float64); bb = a + E1[tri]; cc = a + E2[tri]
        lo = np.minimum(np.minimum(a, bb), cc); hi = np.maximum(np.maximum(a, bb), cc)
        outd = np.maximum(np.maximum(lo - Xh, Xh - hi), 0).max(1)
        dmax = np.abs(d6).max(1)
        ainf = np.maximum(np.abs(o6 - lo), np.abs(o6 - hi)).max(1)
        pgen = np.abs(E1[tri].astype(np.float64)).sum(1) * np.abs(E2[tri].astype(np.float64)).sum(1)
        ut, inv = np.unique(tri, return_inverse=True)
        ps = RA.source_p_probe(A[ut], E1[ut], E2[ut], S_src, sigma, cam).astype(np.float64)[inv]
        slack = 2.0 ** -17 * (ainf + np.abs(o6).max(1))
        rho_g = 216.0 * dmax * ainf * pgen + slack
        rho_s = 216.0 * dmax * ainf * np.where(ainf <= 32.0, ps, pgen) + slack
        cert = ps < pgen * 0.999
        print("%-22s accepted hits %8d on %6d triangles: outside their triangle's box max %.3e (median %.1e); / rho(Pgen) max %.2e; "
              "/ rho(P_S) max %.4f (certificate for %.2f%% of the hits; median rho(P_S) %.2e against rho(Pgen) %.2e)"
              % (name, int(hit.sum()), len(ut), outd.max(), np.median(outd), (outd / rho_g).max(), (outd / rho_s).max(), 100.0 * cert.mean(), np.median(rho_s), np.median(rho_g)), file=out)
        return float((outd / rho_s).max()), float((outd / rho_g).max()), int(hit.sum())

    res = []
    hits, _ = o.probe(np.concatenate([org, d], 1))
    res.append(check("primary rays", org, d, hits, pos.astype(np.float64), 0.0, True))
    # shadow rays as the reference builds them (scene.cpp:787, lights.cpp:32-38): orig = P + N bias, dir = -normalize(P - pos)
    hit = (hits[:, 0] > 0) & (hits[:, 1] == 1)
    t = hits[hit, 3]; tri = hits[hit, 2].astype(np.int64); uu = hits[hit, 4]; vv = hits[hit, 5]
    P = (org[hit] + d[hit] * t[:, None]).astype(f32)
    n = ((NB[tri] * uu[:, None] + NC[tri] * vv[:, None] + NA[tri] * (f32(1) - uu - vv)[:, None]) / f32(3)).astype(f32)
    n = (n * (1.0 / np.sqrt((n.astype(np.float64) ** 2).sum(1))).astype(f32)[:, None]).astype(f32)
    so = (P + n * bias).astype(f32)
    vmax = float(np.abs(tris[:, 0:9]).max())
    for k, lp in enumerate(lights):
        L = (P - lp[None]).astype(f32)
        l2 = (L[:, 0] * L[:, 0] + L[:, 1] * L[:, 1] + L[:, 2] * L[:, 2]).astype(f32)
        L = (L * (1.0 / np.sqrt(l2.astype(np.float64))).astype(f32)[:, None]).astype(f32)
        dist = np.sqrt(l2.astype(np.float64)).astype(f32)
        sd = (-L).astype(f32)
        nb = 1.000001 * float(bias)
        sigma = nb * 1.001 + 4.0 * u24 * (3.01 * (vmax + 32.0 + nb) + 2.01 * float(np.abs(lp).max()))
        # (the line of every shadow ray really passes the light within sigma)
        w = lp[None].astype(np.float64) - so.astype(np.float64); d6 = sd.astype(np.float64)
        perp = w - d6 * ((w * d6).sum(1) / (d6 * d6).sum(1))[:, None]
        assert np.linalg.norm(perp, axis=1).max() <= sigma, (np.linalg.norm(perp, axis=1).max(), sigma)
        sh, _ = o.probe(np.concatenate([so, sd], 1))
        res.append(check("shadow rays, light %d" % k, so, sd, sh, lp.astype(np.float64), sigma, False, limit=dist))
    return res


if __name__ == "__main__":
    stride = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "profiles", "r04_rho_real_data.txt"), "w") as f:
        print("tools/research/rho_real_data.py %d %d: every %d-th row of the headline frame (cfg2_smooth_250k.scene at %dx%d), CPU oracle" % (stride, S, stride, S, S), file=f)
        main(stride, S, f)
    print(open(f.name).read())
