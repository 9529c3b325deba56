"""GPU box experiment: one frame as K row bands pipelined over two streams -- pass 1 of band k+1 overlaps Sobel + SSAA of
band k (the SSAA launches are latency-bound: few, slow waves).  python tools/research/pipeline_exp.py [K ...]
RESULT (round 1): frames are bit-identical to the plain frame, but nothing overlaps -- the persistent pass-1 waves hold
every wave slot until their queue is empty, so the other stream's kernels start when pass 1 ends (rocprofv3 kernel
trace); K = 2 / 4 / 8 bands: 19.5 / 21.4 / 24.5 ms against 17.1 ms.  Overlap needs ONE persistent kernel.
RESULT (round 2, kernels twice as fast): K = 2 / 4 / 8: 9.6 / 10.3 / 11.4 ms against 7.9 ms.  Leaving wave slots free for the
other stream (RTX_PASS1_BLOCKS_PER_CU=5 RTX_SSAA_BLOCKS_PER_CU=1, knobs read at scene creation) makes the two streams
overlap but costs more than it hides: plain 10.1 ms, K = 4 10.2 ms; 5 + 2: 9.4 / 9.7; 4 + 2: 10.3 / 9.8."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import rendering_amd as RA

W = H = 4096
g = RA.Scene("scenes/cfg2_smooth_250k.scene", W, H)
fb = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
mask = torch.zeros((H, W), dtype=torch.uint8, device="cuda")
sp, ss = torch.cuda.Stream(), torch.cuda.Stream()


def plain():
    g.render_pass1(fb); g.sobel(fb, mask); g.render_ssaa(mask, fb)


def pipelined(K):
    edges = [int(round(H * k / K / 64.0)) * 64 for k in range(K + 1)]
    edges[-1] = H
    cur = torch.cuda.current_stream()
    sp.wait_stream(cur); ss.wait_stream(cur)
    done = 0
    for k in range(K):
        a, b = edges[k], edges[k + 1]
        p_end = min(b + 2, H)
        g.render_pass1(fb, rows=(done, p_end), stream=sp)          # rows [done, b + 2): never re-renders a row
        done = p_end
        ev = torch.cuda.Event(); ev.record(sp)
        ss.wait_event(ev)
        g.sobel(fb, mask, rows=(a + (1 if k else 0), min(b + 1, H)), stream=ss)   # mask rows up to the next band's first row
        g.render_ssaa(mask, fb, rows=(a, b), stream=ss)
    cur.wait_stream(sp); cur.wait_stream(ss)


def timeit(fn, n=6):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

plain(); torch.cuda.synchronize(); ref = fb.clone(); refm = mask.clone()
print("plain: %.3f ms/frame" % timeit(plain))
for K in [int(x) for x in sys.argv[1:]] or [2, 4, 8]:
    fb.zero_(); mask.zero_()
    pipelined(K); torch.cuda.synchronize()
    same = bool((fb.view(torch.int32) == ref.view(torch.int32)).all()) and bool((mask == refm).all())
    print("K=%d: %.3f ms/frame, identical to the plain frame: %s" % (K, timeit(lambda: pipelined(K)), same))
