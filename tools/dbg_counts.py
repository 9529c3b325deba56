"""GPU box, RTX_DBG=2 build: wave-level stage counters, certificate outcomes and leaf-size histogram of one
instrumented pass 1 (RTX_DEBUG_ITEMS=1 python tools/dbg_counts.py [scene] [W] [H])."""
import os as _os
_os.environ.setdefault("RTX_ALLOW_ENV_KNOBS", "1")      # (the product ignores RTX_* environment knobs without it)
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import rendering_amd as RA
scene = sys.argv[1] if len(sys.argv) > 1 else "scenes/cfg2_smooth_250k.scene"
W = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
H = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
g = RA.Scene(scene, W, H)
fb = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
g.render_pass1(fb)
torch.cuda.synchronize()
g.counters_enable(not os.environ.get('DBG_PRODUCT'))
g.counters_reset()
g.render_pass1(fb)
torch.cuda.synchronize()
print("instrumented pass1 ms", g.last_kernel_ms(0))
c = g.counters()
print("rays %d box %d tri %d moot %d" % (c[0], c[1], c[2], g.moot_rays))
