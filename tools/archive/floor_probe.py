import os, sys, torch, numpy as np
sys.path.insert(0, os.getcwd())
import rendering_amd as RA
src=open("scenes/cfg2_smooth_250k.scene").read()
nomesh=src[:src.index("[object]\ntype=mesh")]+"[end]\n"
open("/tmp/nomesh.scene","w").write(nomesh)
W=H=4096
fb=torch.zeros((H,W,3),dtype=torch.float32,device="cuda")
for name,path in (("full","scenes/cfg2_smooth_250k.scene"),("floor only","/tmp/nomesh.scene")):
    g=RA.Scene(path,W,H)
    for i in range(4): g.render_pass1(fb)
    torch.cuda.synchronize()
    print(name,"pass1 ms",g.last_kernel_ms(0))
    if name=="full":
        c=g.tile_cost().astype(np.float64)
        # classify tiles by what the centre pixel hit: use fb colours? use cost grid: leaves>0 => mesh region
        refs,leaves=g.cost_grid()
        l=np.repeat(np.repeat(leaves,2,0),2,1)[:c.shape[0],:c.shape[1]]
        m=l>0
        img=fb.cpu().numpy()
        sky=(img[4::8,4::8].sum(-1)[:c.shape[0],:c.shape[1]]==0)
        print("tiles: mesh-box %d (sum ticks %.3g), outside %d (sum %.3g); of the outside: black-centre %d (sum %.3g)"%(m.sum(),c[m].sum(),(~m).sum(),c[~m].sum(),(sky&~m).sum(),c[sky&~m].sum()))
