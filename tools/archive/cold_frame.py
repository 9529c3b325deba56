"""GPU box: the first frame of a fresh scene -- host time of the call, GPU time of the frame, kernels.
python tools/cold_frame.py [scene W H]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import rendering_amd as RA
scene = sys.argv[1] if len(sys.argv) > 1 else "scenes/cfg2_smooth_250k.scene"
W = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
H = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
fb = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda"); mask = torch.zeros((H, W), dtype=torch.uint8, device="cuda")
warm = RA.Scene(scene, W, H); warm.render_frame(fb, mask); torch.cuda.synchronize()      # runtime, code object, caches of the process
for it in range(3):
    g = RA.Scene(scene, W, H); g.gpu(); torch.cuda.synchronize()
    t0 = time.perf_counter(); g.render_frame(fb, mask); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    mode = g.frame_mode()[0]
    ks = [g.last_kernel_ms(k) for k in ((0, 1, 2) if mode == 0 else (4,))]
    t3 = time.perf_counter(); g.render_frame(fb, mask); t4 = time.perf_counter(); torch.cuda.synchronize(); t5 = time.perf_counter()
    print("cold: call returns after %.2f ms, frame done after %.2f ms (events around the frame %.2f ms; kernels %s, %s); second frame: %.2f / %.2f ms"
          % ((t1 - t0) * 1e3, (t2 - t0) * 1e3, g.last_kernel_ms(3), " ".join("%.2f" % k for k in ks), ("three launches", "one launch")[mode], (t4 - t3) * 1e3, (t5 - t3) * 1e3))
    g.close()
