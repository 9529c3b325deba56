#!/usr/bin/env python
"""Headline benchmark: Mrays/s + ms/frame at 4096x4096 on the 250k-triangle BVH scene (BASELINE.json).

One "step" = one frame of the hot path = Scene::launchWorkers (pass 1) + Scene::launchSSAA (Sobel mask +
adaptive 4-ray pass) on synthetic input (scenes/cfg2_smooth_250k.scene, generated mesh), scene resident in HBM,
framebuffer resident in HBM.  A ray = one Render::trace invocation (stats::raysCasted): primary, shadow,
reflect/refract and SSAA rays; rays/frame is counted once by the instrumented kernel variant (deterministic).
Of these, `moot_shadow_rays` are shadow rays whose answer cannot influence the pixel (Diffuse surface turned away from
the light: vis * max(0, N.-L) is +0 either way); the timed kernels do not walk them, the frame is bit-identical.
N > 1: rows are dealt to the ranks in 64-row bands, every frame ends with an RCCL gather to rank 0 (strong
scaling of the same frame).  Prints ONE JSON line on rank 0.
"""
import argparse
import subprocess
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.chdir(ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s


REF_CHILD = r"""
import os, sys, time
sys.path.insert(0, %r)
from tools import ref_harness as R
s = R.RefScene(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), workers=int(sys.argv[4]))
t0 = time.perf_counter(); s.pass1(); dt = time.perf_counter() - t0
print("REF_PASS1_SECONDS %%.6f" %% dt)
"""


def reference_baseline(scene_path, width, height, gpu_scene, cores):
    """The REAL reference (oracle/_ref, built from /root/reference by oracle/Makefile where that tree exists; the .so
    travels with the repo): Scene::launchWorkers of the whole frame with nWorkers = host cores, in a child process
    (the reference prints progress to stdout and keeps process-global option flags).  None if it is not available."""
    from tools import ref_harness
    if not ref_harness.available():
        return None
    try:
        out = subprocess.run([sys.executable, "-c", REF_CHILD % ROOT, scene_path, str(width), str(height), str(cores)],
                             cwd=ROOT, capture_output=True, text=True, timeout=180)
        sec = [float(l.split()[1]) for l in out.stdout.splitlines() if l.startswith("REF_PASS1_SECONDS")]
        if out.returncode != 0 or not sec:
            return None
    except Exception:
        return None
    fb = torch.zeros((height, width, 3), dtype=torch.float32, device="cuda")
    gpu_scene.set_row_ownership(0, 1, 0, False)
    gpu_scene.counters_enable(True)
    gpu_scene.counters_reset()
    gpu_scene.render_pass1(fb)
    rays = int(gpu_scene.counters()[0])
    gpu_scene.counters_enable(False)
    return {"value": round(rays / sec[0] / 1e6, 4), "unit": "Mrays/s", "cores": cores, "kind": "reference",
            "sample": "pass 1 (Scene::launchWorkers, nWorkers = %d) of the whole %dx%d frame by the reference itself: %d rays in %.1f s"
                      % (cores, width, height, rays, sec[0])}


def cpu_baseline(scene_path, width, height, gpu_scene, target_s=15.0):
    """Oracle (CPU restatement: same exhaustive BVH walk, thread per 128x128 tile, all host cores) on a bounded
    sample of the SAME frame: 32-row bands spread evenly over the image, as many as fit ~target_s seconds
    (the whole frame when the host is fast enough).  The rays of the sampled rows are counted by the
    instrumented GPU kernel (tests/test_gpu_parity.py proves those counts identical to the oracle's)."""
    cores = os.cpu_count() or 1
    ref = reference_baseline(scene_path, width, height, gpu_scene, cores) if os.environ.get("BENCH_CPU_BASELINE", "reference") == "reference" else None
    if ref is not None:
        return ref
    from oracle import oracle as O
    o = O.OracleScene(scene_path, width, height)
    band = 32
    all_bands = [(y, min(y + band, height)) for y in range(0, height, band)]
    # probe: every 32nd band
    probe = all_bands[::32]
    ms = 0.0
    for y0, y1 in probe:
        o.pass1(rows=(y0, y1))
        ms += o.last_ms
    est_full = ms * 1e-3 * len(all_bands) / max(len(probe), 1)
    stride = max(1, int(round(est_full / target_s)))
    bands = all_bands[::stride]
    ms = 0.0
    if stride == 1:
        o.pass1()
        ms = o.last_ms
    else:
        for y0, y1 in bands:
            o.pass1(rows=(y0, y1))
            ms += o.last_ms
    fb = torch.zeros((height, width, 3), dtype=torch.float32, device="cuda")
    gpu_scene.set_row_ownership(0, 1, 0, False)
    gpu_scene.counters_enable(True)
    gpu_scene.counters_reset()
    if stride == 1:
        gpu_scene.render_pass1(fb)
    else:
        for y0, y1 in bands:
            gpu_scene.render_pass1(fb, rows=(y0, y1))
    rays = int(gpu_scene.counters()[0])
    gpu_scene.counters_enable(False)
    cores = os.cpu_count() or 1
    what = "whole frame" if stride == 1 else "%d of %d 32-row bands (every %d-th)" % (len(bands), len(all_bands), stride)
    return {"value": round(rays / (ms * 1e-3) / 1e6, 4), "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": "pass 1 (Scene::launchWorkers) of the %dx%d frame, %s: %d rays in %.1f s" % (width, height, what, rays, ms * 1e-3)}


def measured_traffic():
    """HBM bytes per pass-1 launch from a separate rocprofv3 --pmc run of this same command (tools/pmc.sh,
    FETCH_SIZE and WRITE_SIZE in separate passes, KB -> bytes); committed under profiles/.  None if absent."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "r01_pass1_traffic.json")))
        return int(d["fetch_bytes_per_launch"] + d["write_bytes_per_launch"])
    except Exception:
        return None


def valu_issue(avg_ms):
    """Fraction of the VALU issue rate (the kernel's real bound) the pass-1 launch achieves: VALU
    wave-instructions per launch (SQ_INSTS_VALU of a separate rocprofv3 --pmc run of this command, committed under
    profiles/) over what 1024 SIMDs issue in the measured duration at one instruction per 4 cycles, 2.4 GHz."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "r01_pass1_traffic.json")))
        n = float(d["SQ_INSTS_VALU_per_launch"])
        return {"instructions_per_launch": n, "peak_per_s": 1024 * 2.4e9 / 4, "frac": round(n / (avg_ms * 1e-3) / (1024 * 2.4e9 / 4), 4)}
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scene", default="scenes/cfg2_smooth_250k.scene")
    ap.add_argument("--width", type=int, default=4096)
    ap.add_argument("--height", type=int, default=4096)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ssaa", action="store_true")
    ap.add_argument("--verify", action="store_true", help="N > 1: rank 0 also renders the whole frame alone and compares the gathered image with it")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d" % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the product path)")
    # BENCH_DIST_BACKEND=gloo is a functional check of the N > 1 flow on a box with fewer GPUs than ranks: the ranks share
    # the devices and the exchange is staged through host memory (not a measurement; see --verify)
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    rdev = "cuda" if backend == "nccl" else "cpu"          # where the small reductions live
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)

    from rendering_amd import assets, parallel
    import rendering_amd as RA
    if rank == 0:
        assets.ensure(["bumpy_250k.obj"] if "250k" in args.scene else None)
    if world > 1:
        dist.barrier()
    W, H = args.width, args.height
    scene = RA.Scene(args.scene, W, H, device=local)
    fb = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    mask = torch.zeros((H, W), dtype=torch.uint8, device="cuda")
    ssaa = not args.no_ssaa

    img = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda") if world > 1 else None

    def step():
        parallel.shard_frame(scene, fb, mask, world, rank, ssaa=ssaa)
        if world > 1:
            # the frame has to end up in ONE place: quantise to the BGR8 image saveImage writes (4x fewer bytes than
            # the fp32 framebuffer) and send every owned band straight into rank 0's image
            scene.quantize(fb, img)
            if backend == "nccl":
                parallel.gather_frame(img, world, rank, bottom_up=True)
            else:
                host = img.cpu()
                parallel.gather_frame(host, world, rank, bottom_up=True)
                img.copy_(host)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # rays / box tests / triangle tests of one frame under reference semantics (instrumented variant, untimed).
    # Pass 1 is counted over the OWNED rows only (halo off), so that the sum over the ranks is the ray count of the
    # frame itself -- the rays of the recomputed halo rows are extra work of the sharding, not units of the metric.
    scene.counters_enable(True)
    scene.set_row_ownership(parallel.BAND if world > 1 else 0, world, rank, halo=False)
    scene.counters_reset()
    scene.render_pass1(fb)
    c1 = scene.counters()                       # pass 1, this rank's rows
    moot = scene.moot_rays
    scene.counters_enable(False)
    parallel.shard_frame(scene, fb, mask, world, rank, ssaa=False)      # proper framebuffer (with halo rows) for the mask
    scene.counters_enable(True)
    if ssaa:
        scene.counters_reset()
        scene.sobel(fb, mask)
        scene.render_ssaa(mask, fb)
        c2 = scene.counters()
        moot += scene.moot_rays
    else:
        c2 = np.zeros(3, np.int64)
    scene.counters_enable(False)
    tot = torch.tensor([int(x) for x in (c1 + c2)] + [moot], dtype=torch.int64, device=rdev)
    if world > 1:
        dist.all_reduce(tot)
    rays_per_frame = int(tot[0])

    for _ in range(args.warmup):
        step()
    sync()
    scene.kernel_time_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=rdev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax[0])

    verified = None
    if args.verify and world > 1:
        gathered = img.clone()
        if rank == 0:
            parallel.shard_frame(scene, fb, mask, 1, 0, ssaa=ssaa)
            whole = torch.zeros_like(img)
            scene.quantize(fb, whole)
            torch.cuda.synchronize()
            verified = bool((whole == gathered).all())
        sync()
    n1, ms1 = scene.kernel_time_stats(0)
    n2, ms2 = scene.kernel_time_stats(2)
    # algorithmic bytes of ONE pass-1 launch: 32 B per box test + 40 B per triangle test counted
    # under reference traversal semantics + 12 B per rendered pixel
    rendered_px = (W - 1) * (H - 1) / world
    alg_bytes = 32.0 * float(c1[1]) + 40.0 * float(c1[2]) + 12.0 * rendered_px
    avg_ms = ms1 / max(n1, 1)
    achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
    out = {
        "metric": "Mrays/s + ms/frame at 4096^2, 250k-tri BVH scene",
        "value": round(rays_per_frame * args.steps / dt / 1e6, 3),
        "unit": "Mrays/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s @%dx%d, pass 1%s" % (os.path.basename(args.scene), W, H, " + Sobel-adaptive SSAA" if ssaa else ""),
                   "rays_per_frame": rays_per_frame, "moot_shadow_rays": int(tot[3]), "parallelism": "rows in %d-row bands over %d GPU(s)%s" % (parallel.BAND, world, ", BGR8 bands sent to rank 0 over RCCL" if world > 1 else ""),
                   "pass1_ms": round(avg_ms, 3), "ssaa_ms": round(ms2 / max(n2, 1), 3),
                   "pass1_rays_rank0": int(c1[0]), "ssaa_rays_rank0": int(c2[0]), "ssaa_pixels_rank0": int(mask.sum()),
                   "ssaa_box_tests": int(c2[1]), "ssaa_tri_tests": int(c2[2])},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": measured_traffic(),
                     "kernel": "rtxPass1Kernel", "avg_launch_ms": round(avg_ms, 3),
                     "algorithmic_bytes_per_launch": int(alg_bytes),
                     "box_tests": int(c1[1]), "tri_tests": int(c1[2]),
                     "valu_issue": valu_issue(avg_ms) if world == 1 else None},
    }
    if verified is not None:
        out["config"]["gathered_image_equals_single_gpu_image"] = verified
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.scene, W, H, scene)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
