import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import rendering_amd as RA
from rendering_amd import assets
assets.ensure(["bumpy_250k.obj"])
for W, H in ((1920, 1080), (4096, 4096)):
    g = RA.Scene("scenes/cfg2_smooth_250k.scene", W, H)
    fb = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    for _ in range(3): g.render_pass1(fb)
    torch.cuda.synchronize()
    c = g.tile_cost().astype(np.float64) * 1e-5
    cx, cy = (W // 2 - 1) // 8, (H // 2 - 1) // 8          # the tile column / row holding the pixels whose rays have dir.x / dir.y == 0
    ys = slice(int(c.shape[0] * 0.3), int(c.shape[0] * 0.7))
    print("%dx%d pass1 %.3f ms: tile column %d (dir.x = 0 inside) mean %.4f ms over the mesh's rows; its neighbours %d: %.4f, %d: %.4f; slowest tile of the frame (%d,%d) = %.3f ms, slowest in column %d = %.3f, in column %d = %.3f"
          % (W, H, g.last_kernel_ms(0), cx, c[ys, cx].mean(), cx - 1, c[ys, cx - 1].mean(), cx + 1, c[ys, cx + 1].mean(),
             int(np.argmax(c) % c.shape[1]), int(np.argmax(c) // c.shape[1]), c.max(), cx, c[:, cx].max(), cx + 1, c[:, cx + 1].max()))
    g.close()
