"""GPU box: frames of one view back to back, serial (one stream: pass 1, Sobel, SSAA, next frame) against pipelined (two framebuffers; pass 1 + Sobel of frame k + 1 on
stream A beside the SSAA launch of frame k on stream B).  Every frame is compared with the serial one bit for bit.  python tools/pipeline_probe.py [config] [frames]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); os.chdir(ROOT)
import numpy as np, torch
import rendering_amd as RA
from rendering_amd import assets
import bench

cfg = sys.argv[1] if len(sys.argv) > 1 else "headline"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 40
path, W, H = bench.CONFIGS[cfg]
assets.ensure(["bumpy_250k.obj"] if "250k" in path else None)
g = RA.Scene(path, W, H); g.gpu()
g.set_frame_mode(0)
fb = [torch.zeros((H, W, 3), dtype=torch.float32, device="cuda") for _ in range(2)]
mask = [torch.zeros((H, W), dtype=torch.uint8, device="cuda") for _ in range(2)]
for _ in range(4):
    g.render_frame(fb[0], mask[0])
torch.cuda.synchronize()
ref = fb[0].clone(); refm = mask[0].clone()
t0 = time.perf_counter()
for k in range(K):
    g.render_frame(fb[0], mask[0])
torch.cuda.synchronize()
serial = (time.perf_counter() - t0) / K * 1e3
PRI = int(os.environ.get("PIPE_PRIO", "0"))
A, B = torch.cuda.Stream(), torch.cuda.Stream(priority=-1 if PRI else 0)
done_ssaa = [None, None]
def frame(k):
    i = k & 1
    if done_ssaa[i] is not None:
        A.wait_event(done_ssaa[i])          # frame k - 2 has left this framebuffer
    g.render_pass1(fb[i], stream=A)
    g.sobel(fb[i], mask[i], stream=A)
    e = torch.cuda.Event(); e.record(A)
    B.wait_event(e)
    g.render_ssaa(mask[i], fb[i], stream=B)
    d = torch.cuda.Event(); d.record(B); done_ssaa[i] = d
for k in range(6):
    frame(k)
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(K):
    frame(k)
torch.cuda.synchronize()
piped = (time.perf_counter() - t0) / K * 1e3
ok = all(bool((fb[i] == ref).all()) and bool((mask[i] == refm).all()) for i in range(2))
print("%s: serial %.3f ms/frame, pipelined (SSAA of frame k beside pass 1 of frame k + 1) %.3f ms/frame, frames identical: %s" % (cfg, serial, piped, ok))
