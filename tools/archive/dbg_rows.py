"""GPU box, RTX_DBG build: wave-level counters of pass 1 restricted to image rows [y0, y1) (product variant unless
DBG_STATS=1).  RTX_DEBUG_ITEMS=1 python tools/dbg_rows.py y0 y1 [scene] [W] [H]"""
import os as _os
_os.environ.setdefault("RTX_ALLOW_ENV_KNOBS", "1")      # (the product ignores RTX_* environment knobs without it)
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import rendering_amd as RA
y0, y1 = int(sys.argv[1]), int(sys.argv[2])
scene = sys.argv[3] if len(sys.argv) > 3 else "scenes/cfg2_smooth_250k.scene"
W = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
H = int(sys.argv[5]) if len(sys.argv) > 5 else 4096
g = RA.Scene(scene, W, H)
fb = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
g.render_pass1(fb, rows=(y0, y1))
torch.cuda.synchronize()
g.counters_enable(bool(os.environ.get("DBG_STATS")))
g.counters_reset()
g.render_pass1(fb, rows=(y0, y1))
torch.cuda.synchronize()
print("rows %d..%d: pass1 %.3f ms" % (y0, y1, g.last_kernel_ms(0)))
g.counters()
c = g.tile_cost()[y0 // 8]
import numpy as np
k = np.argsort(c)[::-1][:6]
print("slowest tiles of the row:", [(int(x), round(float(c[x]) * 1e-5, 3)) for x in k])
