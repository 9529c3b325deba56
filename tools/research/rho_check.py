"""How far from its triangle can a hit lie that the reference ACCEPTS (objects.cpp:59-95 in fp32)?  The prune records of the wide
walk rest on the bound  |orig + t_c dir - triangle box|_inf <= 36 u dmax ainf s1 s2 / det_c  (rtx_kernels.hip,
pruneAlive).  Random and adversarial (ray, triangle) pairs -- rays nearly in the plane of the triangle, slivers -- in the
reference's arithmetic (numpy fp32, no FMA); reports the worst observed distance over the bound.
python tools/research/rho_check.py [seed]      (tests/test_prune_bound_cpu.py runs a smaller sample)"""
import sys
import numpy as np
f32=np.float32
u=2.0**-24
def run(N, mode, rng, quiet=False):
    # triangles
    sc = 10.0**rng.uniform(-3,0,(N,1))
    v0 = rng.uniform(-1,1,(N,3))*10.0**rng.uniform(-1,0.5,(N,1))
    e1 = rng.normal(size=(N,3))*sc; e2 = rng.normal(size=(N,3))*sc*10.0**rng.uniform(-2,0,(N,1))
    if mode=='sliver':
        e2 = e1*rng.uniform(0.2,1.5,(N,1)) + rng.normal(size=(N,3))*sc*1e-3
    v0=v0.astype(f32); e1=e1.astype(f32); e2=e2.astype(f32)
    n = np.cross(e1.astype(np.float64), e2.astype(np.float64)); nn = n/np.linalg.norm(n,axis=1)[:,None]
    # target point in/near the triangle
    a_,b_ = rng.uniform(-0.3,1.3,(2,N,1))
    tgt = v0 + a_*e1 + b_*e2
    # direction nearly in plane
    tang = e1.astype(np.float64)*rng.normal(size=(N,1)) + e2.astype(np.float64)*rng.normal(size=(N,1)); tang/=np.linalg.norm(tang,axis=1)[:,None]
    ang = 10.0**rng.uniform(-7,-0.5,(N,1))
    if mode=='generic': ang = rng.uniform(0.05,1.5,(N,1))
    d = tang*np.cos(ang) - nn*np.sin(ang)   # facing: d.n<0 -> det>0?  sign both ways below
    d *= rng.choice([-1,1],(N,1))*0+1
    dist = 10.0**rng.uniform(-2,1,(N,1))
    o = tgt - d*dist
    o=o.astype(f32); d=d.astype(f32)
    # reference fp32
    px = d[:,1]*e2[:,2]-d[:,2]*e2[:,1]; py=d[:,2]*e2[:,0]-d[:,0]*e2[:,2]; pz=d[:,0]*e2[:,1]-d[:,1]*e2[:,0]
    det = e1[:,0]*px+e1[:,1]*py+e1[:,2]*pz
    ok = ~(det.astype(np.float64)<1e-8)
    with np.errstate(all='ignore'):
        inv=f32(1)/det
        tx=o[:,0]-v0[:,0]; ty=o[:,1]-v0[:,1]; tz=o[:,2]-v0[:,2]
        uu=(tx*px+ty*py+tz*pz)*inv
        ok&=~((uu<0)|(uu>1))
        qx=ty*e1[:,2]-tz*e1[:,1]; qy=tz*e1[:,0]-tx*e1[:,2]; qz=tx*e1[:,1]-ty*e1[:,0]
        vv=(d[:,0]*qx+d[:,1]*qy+d[:,2]*qz)*inv
        ok&=~((vv<0)|(uu+vv>1))
        tt=(e2[:,0]*qx+e2[:,1]*qy+e2[:,2]*qz)*inv
        ok&=~(tt<0)
    idx=np.nonzero(ok)[0]
    o6=o[idx].astype(np.float64); d6=d[idx].astype(np.float64); t6=tt[idx].astype(np.float64)
    P=o6+t6[:,None]*d6
    A=v0[idx].astype(np.float64); B=A+e1[idx].astype(np.float64); Cc=A+e2[idx].astype(np.float64)
    lo=np.minimum(np.minimum(A,B),Cc); hi=np.maximum(np.maximum(A,B),Cc)
    out=np.maximum(np.maximum(lo-P,P-hi),0).max(1)
    s1=np.abs(e1[idx].astype(np.float64)).sum(1); s2=np.abs(e2[idx].astype(np.float64)).sum(1)
    dmax=np.abs(d6).max(1); ainf=np.abs(o6-A).max(1)
    rho=35.6*u*dmax*ainf*s1*s2/det[idx].astype(np.float64) + 4*u*(np.abs(P).max(1))
    rho_unc=35.6*u*dmax*ainf*s1*s2/1e-8
    r=out/rho
    k=np.argmax(r) if len(r) else 0
    if not quiet: print(mode, "accepted %d of %d; max outside/rho = %.3f (outside %.3e rho %.3e det %.2e), max outside/rho_unc %.3f; frac with outside>0: %.3f" % (len(idx), N, r.max() if len(r) else 0, out[k] if len(r) else 0, rho[k] if len(r) else 0, det[idx][k] if len(r) else 0, (out/rho_unc).max() if len(r) else 0, (out>0).mean() if len(r) else 0))
    return (float(r.max()) if len(r) else 0.0), len(idx), (float(out.max()) if len(r) else 0.0)

if __name__ == "__main__":
    rng=np.random.default_rng(int(sys.argv[1]) if len(sys.argv)>1 else 0)
    for mode in ['graze','sliver','generic']:
        run(2000000, mode, rng)
