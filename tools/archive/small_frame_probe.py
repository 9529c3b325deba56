"""GPU box: kernel times of small frames (latency floor).  python tools/small_frame_probe.py [scene ...]"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import rendering_amd as RA
scenes = sys.argv[1:] or ["scenes/cfg1_simple_shapes.scene", "scenes/cfg3_reflective_refractive.scene"]
for scene in scenes:
    for W, H in ((64, 64), (256, 256), (512, 512), (1024, 1024), (1920, 1080)):
        g = RA.Scene(scene, W, H)
        fb = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
        mask = torch.zeros((H, W), dtype=torch.uint8, device="cuda")
        best = [1e9, 1e9, 1e9]
        for it in range(6):
            g.render_pass1(fb); g.sobel(fb, mask); g.render_ssaa(mask, fb)
            torch.cuda.synchronize()
            if it >= 2:
                for k in range(3):
                    best[k] = min(best[k], g.last_kernel_ms(k))
        print("%s %dx%d: pass1 %.3f sobel %.3f ssaa %.3f ms, flagged %d" % (os.path.basename(scene), W, H, best[0], best[1], best[2], int(mask.sum())))
