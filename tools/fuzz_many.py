"""GPU box, one-off: many more random scenes than the test suite holds (tests/test_gpu_fuzz.py: make_scene), plus random
cameras / rotations / scales of the 250k and 25k meshes at larger frames.  python tools/fuzz_many.py [first seed] [n]"""
import os, random, sys, tempfile
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
os.chdir(ROOT)
import torch
import rendering_amd as RA
from rendering_amd import assets
from oracle import oracle as O
from tests.test_gpu_fuzz import make_scene, bits
from tests.util_rays import probe_rays

assets.ensure(); assets.ensure(["bumpy_250k.obj"])
first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
bad = 0
tmp = tempfile.mkdtemp()
for seed in range(first, first + n):
    r = random.Random(seed)
    if seed % 5 == 4:
        # a big mesh seen from a random place, random rotation / scale / culling, SSAA on
        w, h = r.choice([(256, 192), (320, 200), (192, 256)])
        mesh = r.choice(["bumpy_250k.obj", "bumpy_25k.obj", "bumpy_25k.obj"])
        s = "[options]\nwidth=%d\nheight=%d\nfov=%d\nposition=%.3f,%.3f,%.3f\nrotation=%.2f,%.2f,%.2f\nuseBackfaceCulling=%d\nimage_name=output/fuzz\n\n" % (
            w, h, r.choice([30, 60, 90]), r.uniform(-1, 1), r.uniform(-1, 1), r.uniform(-0.5, 1.5), r.uniform(-20, 20), r.uniform(-20, 20), r.uniform(-30, 30), r.randrange(2))
        s += "[light]\ntype=point\nposition=%.2f,%.2f,%.2f\ncolor=1,1,1\nintensity=1.5\n\n[light]\ntype=distant\ndirection=%.2f,%.2f,%.2f\ncolor=0.5,0.7,1\nintensity=0.4\n\n" % (
            r.uniform(-2, 2), r.uniform(0, 3), r.uniform(-2, 1), r.uniform(-1, 1), r.uniform(-1, -0.2), r.uniform(-1, 0))
        s += "[object]\ntype=plane\npos=0,-1.6,0\nnormal=0,1,0\ncolor=1,1,1\n\n[object]\ntype=mesh\npos=%.3f,%.3f,%.3f\nsize=%.2f,%.2f,%.2f\nrot=%.1f,%.1f,%.1f\ncolor=1,1,1\n%sname=scenes/assets/%s\n\n[end]\n" % (
            r.uniform(-0.5, 0.5), r.uniform(-0.3, 0.3), r.uniform(-4, -2.5), *(r.uniform(1.2, 2.5) for _ in range(3)), r.uniform(-90, 90), r.uniform(-90, 90), r.uniform(-90, 90),
            r.choice(["", "", "material=reflective\n", "material=phong,0.3,0.4,0.6,10\n"]), mesh)
    else:
        w, h = 96 + 8 * (seed % 3), 72 + 4 * (seed % 5)
        s = make_scene(seed, w, h)
    path = os.path.join(tmp, "fuzz%d.scene" % seed)
    open(path, "w").write(s)
    o = O.OracleScene(path, w, h); g = RA.Scene(path, w, h)
    ref1 = o.pass1(); got1 = g.render_host(ssaa=False)
    ref2 = o.ssaa(ref1); got2 = g.render_host(ssaa=True)
    rh, rc = o.probe(probe_rays(256)); gh, gc = g.cast_rays(probe_rays(256))
    ok = np.array_equal(bits(ref1), bits(got1)) and np.array_equal(bits(ref2), bits(got2)) and np.array_equal(bits(rh), bits(gh)) and np.array_equal(bits(rc), bits(gc))
    # rtx_render_frame: three launches and one launch, a cold frame and two warm ones (warm frames split their slow tiles)
    fb = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda"); mask = torch.zeros((h, w), dtype=torch.uint8, device="cuda")
    frames = 0
    for mode in (0, 1):
        g.set_frame_mode(mode)
        for it in range(3):
            fb.zero_(); g.render_frame(fb, mask); g.frame_status()
            if not np.array_equal(bits(ref2), bits(fb.cpu().numpy())):
                frames += 1
    ok = ok and frames == 0
    if not ok:
        bad += 1
        if frames: print("   rtx_render_frame differs in %d of 6 frames" % frames)
        print("MISMATCH seed", seed, "pass-1 pixels", int((bits(ref1) != bits(got1)).any(-1).sum()), "ssaa pixels", int((bits(ref2) != bits(got2)).any(-1).sum()))
    o.close(); g.close()
    if (seed - first) % 250 == 249:      # (a run cut short by its time limit still leaves a record)
        print("progress: seeds %d..%d: %d mismatching scenes" % (first, seed, bad), flush=True)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from srchash import source_hash
print("seeds %d..%d: %d mismatching scenes; sources %s" % (first, first + n - 1, bad, source_hash()))
