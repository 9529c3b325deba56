"""GPU box, build with -DRTX_WAVE_TRACE=1 (a product build + three timestamps per wave): how the waves of a pass 1 end -- the launch lasts as long as its last wave.
RTX_DEBUG_ITEMS=1 python tools/wave_tail.py [scene W H [parts part band]]      (parts: one device's share of the frame, rtx_set_row_ownership)"""
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
if len(sys.argv) > 6:
    g.set_row_ownership(int(sys.argv[6]), int(sys.argv[4]), int(sys.argv[5]), True)
for _ in range(4):
    g.render_pass1(fb)
torch.cuda.synchronize()
print("pass1 ms", g.last_kernel_ms(0))
g.counters()
