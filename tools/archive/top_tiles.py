"""GPU box: the slowest pass-1 tiles of a view.  python tools/top_tiles.py scene W H [n] -> 'tx,ty' per line (stdout), costs on stderr"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import rendering_amd as RA
scene, W, H = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
n = int(sys.argv[4]) if len(sys.argv) > 4 else 3
g = RA.Scene(scene, W, H)
fb = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
g.render_pass1(fb); g.render_pass1(fb); torch.cuda.synchronize()
c = g.tile_cost()
idx = np.argsort(-c.ravel())[:n]
for i in idx:
    ty, tx = divmod(int(i), c.shape[1])
    print("%d,%d" % (tx, ty))
    print("tile %d,%d: %.3f ms" % (tx, ty, c[ty, tx] * 1e-5), file=sys.stderr)
print("pass 1 %.3f ms; tiles over 0.5 ms: %d, over 1 ms: %d" % (g.last_kernel_ms(0), (c > 50000).sum(), (c > 100000).sum()), file=sys.stderr)
