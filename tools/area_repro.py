"""GPU box: is pass 1 of a scene deterministic and equal to the reference's?  python tools/area_repro.py [scene W H runs]"""
import os, sys, subprocess, json, tempfile
import numpy as np, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import rendering_amd as RA
import bench
scene = sys.argv[1] if len(sys.argv) > 1 else "scenes/area_light.scene"
W = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
H = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
runs = int(sys.argv[4]) if len(sys.argv) > 4 else 12
g = RA.Scene(scene, W, H)
fb = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
# the reference's own pass 1 (oracle/_ref, as bench.py runs it)
dump = os.path.join(tempfile.gettempdir(), "repro_ref")
out = subprocess.run([sys.executable, "-c", bench.REF_CHILD % ROOT, scene, str(W), str(H), str(os.cpu_count()), dump, "0", "{}"], cwd=ROOT, capture_output=True, text=True, timeout=900)
ref = np.load(dump + ".pass1.npy") if out.returncode == 0 and os.path.exists(dump + ".pass1.npy") else None
print("reference frame:", None if ref is None else ref.shape, out.stderr[-200:] if ref is None else "")
first = None
for k in range(runs):
    if k == runs // 2:
        g.counters_enable(True); g.counters_reset(); g.render_pass1(fb); g.counters_enable(False)      # (what bench.py does in between)
    fb.zero_()
    g.render_pass1(fb); torch.cuda.synchronize()
    got = fb.cpu().numpy()
    if first is None: first = got.copy()
    d0 = (got.view(np.uint32) != first.view(np.uint32)).any(-1)
    msg = "run %2d: differs from run 0 in %d pixels" % (k, int(d0.sum()))
    if ref is not None:
        d1 = (got.view(np.uint32) != ref.view(np.uint32)).any(-1)
        msg += ", from the reference in %d" % int(d1.sum())
        for (y, x) in list(zip(*np.nonzero(d1)))[:4]:
            msg += " | (%d,%d) gpu %s ref %s" % (x, y, got[y, x].tolist(), ref[y, x].tolist())
    print(msg)
