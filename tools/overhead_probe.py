"""GPU box: what a trace round costs OUTSIDE the walk -- pass 1 of the headline frame with the mesh moved behind the camera (same
kernel, same floor, every ray fails the mesh's root box) against the real frame.  Under rocprofv3 --pmc SQ_INSTS_VALU the two
launches give the instruction counts.  python tools/overhead_probe.py"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
os.chdir(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import rendering_amd as RA
src = open("scenes/cfg2_smooth_250k.scene").read()
open("/tmp/behind.scene", "w").write(src.replace("pos=0,0,-3", "pos=0,0,300"))
W = H = 4096
fb = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
for name, path in (("real frame", "scenes/cfg2_smooth_250k.scene"), ("mesh behind the camera", "/tmp/behind.scene")):
    g = RA.Scene(path, W, H)
    best = 1e9
    for i in range(40):          # (long enough for the clocks to come back after the idle time of the scene load)
        g.render_pass1(fb)
        if i >= 30:
            torch.cuda.synchronize(); best = min(best, g.last_kernel_ms(0))
    print(name, "pass1 ms %.3f" % best)
    g.counters_enable(True); g.counters_reset(); g.render_pass1(fb); torch.cuda.synchronize()
    print("   rays", int(g.counters()[0])); g.counters_enable(False)
