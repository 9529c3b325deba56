"""GPU box, product build: how long the SSAA work items of a frame took (per tile: its slowest item, scaled to 16 pixels).
python tools/ssaa_items.py [scene W H]"""
import os, sys, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import rendering_amd as RA
from rendering_amd import _np_ptr
scene = sys.argv[1] if len(sys.argv) > 1 else "scenes/cfg2_smooth_250k.scene"
W = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
H = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
g = RA.Scene(scene, W, H)
fb = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda"); mask = torch.zeros((H, W), dtype=torch.uint8, device="cuda")
g.set_frame_mode(0)
for it in range(4): g.render_frame(fb, mask)
torch.cuda.synchronize()
ty, tx = (H + 7) // 8, (W + 7) // 8
out = np.zeros((2, ty, tx), np.uint32)
RA._check(g.rtx.rtx_tile_cost_read(g.gpu(), _np_ptr(out), out.size), "rtx_tile_cost_read")
p1, it = out[0] * 1e-5, out[1] * 1e-5          # ms
print("pass 1 %.3f ms, SSAA stage %.3f ms" % (g.last_kernel_ms(0), g.last_kernel_ms(2)))
fl = mask.cpu().numpy().reshape(ty, 8, tx, 8).sum((1, 3)) if H % 8 == 0 and W % 8 == 0 else None
nz = it[it > 0]
print("tiles with SSAA items %d; slowest item of a tile (ms): " % len(nz) + ", ".join("%g%% %.3f" % (q, np.percentile(nz, q)) for q in (50, 90, 99, 99.9, 100)))
if fl is not None:
    items = np.ceil(fl / 16.0)
    print("items (16 px) %d; sum over tiles of items x slowest item = %.1f wave-ms (%.3f ms over 4096 waves)" % (items.sum(), (items * it).sum(), (items * it).sum() / 4096))
order = np.argsort(-it.ravel())[:12]
for o in order:
    y, x = divmod(int(o), tx)
    print("  tile (%d, %d) px (%d, %d): SSAA item %.3f ms, pass 1 %.3f ms, flagged %s" % (x, y, x * 8, y * 8, it[y, x], p1[y, x], "?" if fl is None else int(fl[y, x])))
heavy = p1 > 0.25
print("tiles whose pass 1 took > 0.25 ms: %d with SSAA items (slowest %.3f ms); the others: %d (slowest %.3f ms)" % ((heavy & (it > 0)).sum(), it[heavy].max() if heavy.any() else 0, (~heavy & (it > 0)).sum(), it[~heavy].max()))
