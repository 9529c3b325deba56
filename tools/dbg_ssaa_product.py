"""GPU box, RTX_DBG=1 build: wave-level counters of the PRODUCT SSAA launch (walks, node visits, leaves, passes, exact tests, rounds) at the headline.
python tools/dbg_ssaa_product.py [scene] [W] [H]"""
import os as _os
_os.environ.setdefault("RTX_ALLOW_ENV_KNOBS", "1")
_os.environ.setdefault("RTX_DEBUG_ITEMS", "1")
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
print("flagged %d" % int(mask.sum()))
sys.stderr.write("== counters of pass 1 + ssaa so far (ignore)\n"); g.counters()
g.counters_reset()
g.render_ssaa(mask, fb0)
torch.cuda.synchronize()
sys.stderr.write("== product SSAA launch alone (RTX_DBG build) %.3f ms\n" % g.last_kernel_ms(2))
g.counters()
