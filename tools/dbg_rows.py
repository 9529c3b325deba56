"""GPU box, RTX_DBG build: wave-level counters of pass 1 over a few row bands.  RTX_DEBUG_ITEMS=1 python tools/dbg_rows.py y0 y1 [y0 y1 ...]"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import rendering_amd as RA
g = RA.Scene("scenes/cfg2_smooth_250k.scene", 4096, 4096)
fb = torch.zeros((4096, 4096, 3), dtype=torch.float32, device="cuda")
g.counters_enable(True)
a = [int(x) for x in sys.argv[1:]]
for y0, y1 in zip(a[::2], a[1::2]):
    g.counters_reset()
    g.render_pass1(fb, rows=(y0, y1))
    torch.cuda.synchronize()
    sys.stderr.write("rows %d..%d: " % (y0, y1)); sys.stderr.flush()
    c = g.counters()
    print("rows", y0, y1, "rays", c[0], "box", c[1], "tri", c[2], "ms", g.last_kernel_ms(0))
