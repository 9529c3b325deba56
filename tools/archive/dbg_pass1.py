"""GPU box, RTX_DBG build: per-wave timeline summary of one pass 1 (RTX_DEBUG_ITEMS=1 python tools/dbg_pass1.py)."""
import os as _os
_os.environ.setdefault("RTX_ALLOW_ENV_KNOBS", "1")      # (the product ignores RTX_* environment knobs without it)
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import rendering_amd as RA
g = RA.Scene("scenes/cfg2_smooth_250k.scene", 4096, 4096)
fb = torch.zeros((4096, 4096, 3), dtype=torch.float32, device="cuda")
g.render_pass1(fb); g.render_pass1(fb); g.render_pass1(fb)
torch.cuda.synchronize()
print("pass1 ms", g.last_kernel_ms(0))
g.counters()
