"""GPU box: rtx_render_frame against the three separate launches (bit-exact fb + mask) and its time.
python tools/frame_check.py [scene W H] ..."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import rendering_amd as RA
cases = []
a = sys.argv[1:]
while len(a) >= 3:
    cases.append((a[0], int(a[1]), int(a[2]))); a = a[3:]
if not cases:
    cases = [("scenes/cfg1_simple_shapes.scene", 64, 64), ("scenes/cfg1_simple_shapes.scene", 512, 512), ("scenes/cfg2_smooth_4k.scene", 200, 120),
             ("scenes/cfg2_smooth_250k.scene", 1920, 1080), ("scenes/cfg2_smooth_250k.scene", 4096, 4096)]
for scene, W, H in cases:
    g = RA.Scene(scene, W, H)
    fb = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda"); mask = torch.zeros((H, W), dtype=torch.uint8, device="cuda")
    fb2 = torch.zeros_like(fb); mask2 = torch.full_like(mask, 7)
    for it in range(3):
        fb.zero_(); g.render_pass1(fb); g.sobel(fb, mask); g.render_ssaa(mask, fb)
    torch.cuda.synchronize()
    split = g.last_kernel_ms(0) + g.last_kernel_ms(1) + g.last_kernel_ms(2)
    best = 1e9
    modes = ""
    for it in range(8):
        fb2.zero_(); g.render_frame(fb2, mask2)
        g.frame_status()
        best = min(best, g.last_kernel_ms(3)) if it >= 4 else best
        modes += "SF"[g.frame_mode()[0]]
    same = torch.equal(fb.view(torch.int32), fb2.view(torch.int32)); msame = torch.equal(mask, mask2)
    print("%s %dx%d: split %.3f ms (pass1 %.3f ssaa %.3f), render_frame %.3f ms (modes %s, measured split %.3f fused %.3f), fb identical %s, mask identical %s, flagged %d"
          % (os.path.basename(scene), W, H, split, g.last_kernel_ms(0), g.last_kernel_ms(2), best, modes, g.frame_mode()[1], g.frame_mode()[2], same, msame, int(mask.sum())), flush=True)
    if not same:
        d = (fb.view(torch.int32) != fb2.view(torch.int32)).any(dim=2)
        ys, xs = torch.nonzero(d, as_tuple=True)
        print("   differing pixels %d, first (x,y) %s, flagged there %s" % (int(d.sum()), [(int(xs[i]), int(ys[i])) for i in range(min(5, len(xs)))], [int(mask[ys[i], xs[i]]) for i in range(min(5, len(xs)))]))
