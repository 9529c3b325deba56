"""GPU box: slowest SSAA work item and the sum over the items (instrumented variant) next to the product launch time.
RTX_DEBUG_ITEMS=1 python tools/dbg_ssaa.py [scene] [W] [H]"""
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
mask = torch.zeros((H, W), dtype=torch.uint8, device="cuda")
for it in range(2):
    fb.zero_(); g.render_pass1(fb); g.sobel(fb, mask); fb0 = fb.clone(); g.render_ssaa(mask, fb)
torch.cuda.synchronize()
print("product: pass1 %.3f ms, ssaa %.3f ms, flagged %d" % (g.last_kernel_ms(0), g.last_kernel_ms(2), int(mask.sum())))
g.counters_enable(True); g.counters_reset()
g.render_ssaa(mask, fb0)
torch.cuda.synchronize()
print("instrumented ssaa %.3f ms" % g.last_kernel_ms(2))
c = g.counters()
print("rays %d box %d tri %d" % tuple(c))
