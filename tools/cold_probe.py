import os, sys, time
import torch
sys.path.insert(0, os.getcwd())
import rendering_amd as RA
scene="scenes/cfg2_smooth_250k.scene"; W=H=4096
fb = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda"); mask = torch.zeros((H, W), dtype=torch.uint8, device="cuda")
fb2 = torch.zeros_like(fb); mask2=torch.zeros_like(mask)
warm = RA.Scene(scene, W, H); warm.set_frame_mode(0)
for i in range(4): warm.render_frame(fb2, mask2)
torch.cuda.synchronize()
print("warm pass1", warm.last_kernel_ms(0))
for busy in (0,1,1,0):
    g = RA.Scene(scene, W, H); g.gpu(); g.set_frame_mode(0); torch.cuda.synchronize()
    if busy:
        for i in range(3): warm.render_frame(fb2, mask2)
    g.render_frame(fb, mask); torch.cuda.synchronize()
    a=g.last_kernel_ms(0)
    g.render_frame(fb, mask); torch.cuda.synchronize()
    b=g.last_kernel_ms(0)
    g.render_frame(fb, mask); torch.cuda.synchronize()
    print("busy-before=%d: cold pass1 %.2f ms, second %.2f, third %.2f"%(busy,a,b,g.last_kernel_ms(0)))
    g.close()
