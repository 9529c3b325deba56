#!/usr/bin/env python
"""Headline benchmark: Mrays/s + ms/frame at 4096x4096 on the 250k-triangle BVH scene (BASELINE.json).

One "step" = one frame of the hot path = Scene::launchWorkers (pass 1) + Scene::launchSSAA (Sobel mask +
adaptive 4-ray pass) on synthetic input (scenes/cfg2_smooth_250k.scene, generated mesh), scene resident in HBM,
framebuffer resident in HBM.  A ray = one Render::trace invocation (stats::raysCasted): primary, shadow,
reflect/refract and SSAA rays; rays/frame is counted once by the instrumented kernel variant (deterministic).
Of these, `moot_shadow_rays` are shadow rays whose answer cannot influence the pixel (Diffuse surface turned away from
the light: vis * max(0, N.-L) is +0 either way); the timed kernels do not walk them, the frame is bit-identical.
N > 1: rows are dealt to the ranks in bands of 64 to 256 rows (rendering_amd/parallel.py band_height), every frame ends
with an RCCL gather to rank 0 (rtx_gather; strong scaling of the same frame), and after the timed region the gathered
image is compared with the frame rank 0 renders alone (--no-verify skips that).  Prints ONE JSON line on rank 0.
"""
import argparse
import subprocess
import tempfile
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
import numpy as np
s = R.RefScene(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), workers=int(sys.argv[4]))
import json
for k, v in json.loads(sys.argv[7]).items():      # options:: flags of the workload (process-global in the reference: set after the load)
    R.lib().ref_set_flag(k.encode(), int(v))
t0 = time.perf_counter(); fb = s.pass1(); dt = time.perf_counter() - t0
print("REF_PASS1_SECONDS %%.6f" %% dt)
# (untimed) the framebuffers themselves, for the whole-frame comparison with what the GPU renders in the timed region
np.save(sys.argv[5] + ".pass1.npy", fb)
if int(sys.argv[6]):
    t0 = time.perf_counter(); fb2 = s.ssaa(fb); dt = time.perf_counter() - t0
    print("REF_SSAA_SECONDS %%.6f" %% dt)
    np.save(sys.argv[5] + ".frame.npy", fb2)
"""


def reference_baseline(scene_path, width, height, gpu_scene, cores, ssaa=True, flags=None):
    """The REAL reference (oracle/_ref, built from /root/reference by oracle/Makefile where that tree exists; the .so
    travels with the repo): Scene::launchWorkers of the whole frame with nWorkers = host cores, in a child process
    (the reference prints progress to stdout and keeps process-global option flags).  None if it is not available."""
    from tools import ref_harness
    if not ref_harness.available():
        return None
    try:
        dump = os.path.join(tempfile.gettempdir(), "bench_ref_%d" % os.getpid())
        out = subprocess.run([sys.executable, "-c", REF_CHILD % ROOT, scene_path, str(width), str(height), str(cores), dump, "1" if ssaa else "0", json.dumps(flags or {})],
                             cwd=ROOT, capture_output=True, text=True, timeout=900)
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
    res = {"value": round(rays / sec[0] / 1e6, 4), "unit": "Mrays/s", "cores": cores, "kind": "reference",
           "sample": "pass 1 (Scene::launchWorkers, nWorkers = %d) of the whole %dx%d frame by the reference itself: %d rays in %.1f s"
                     % (cores, width, height, rays, sec[0])}
    # Whole-frame parity of what is timed: the reference's own framebuffers of THIS frame against the product path's
    # (rtx_render_frame, as in the timed region), bit for bit -- pass 1 everywhere; the frame after SSAA everywhere but row 0 / column 0,
    # where the reference reads uninitialised mask entries.
    try:
        ref1 = np.load(dump + ".pass1.npy")
        gpu_scene.render_pass1(fb)
        torch.cuda.synchronize()
        got = fb.cpu().numpy()
        nd = int((got.view(np.uint32) != ref1.view(np.uint32)).any(-1).sum())
        res["parity"] = {}
        if nd != 0:
            # The reference is not always equal to ITSELF: AreaLight::setPoints (lights.cpp:46-63) fills the light's sample points lazily, from whichever
            # worker thread needs them first, while the others already read them -- with 256 workers the first pixels of some of their tiles come out NaN
            # now and then.  So a differing pixel is looked at again in a second run of the reference: where its two runs disagree with each other, the
            # product has to equal one of them; everywhere else, both.
            out2 = subprocess.run([sys.executable, "-c", REF_CHILD % ROOT, scene_path, str(width), str(height), str(cores), dump + "b", "0", json.dumps(flags or {})],
                                  cwd=ROOT, capture_output=True, text=True, timeout=900)
            if out2.returncode == 0 and os.path.exists(dump + "b.pass1.npy"):
                ref1b = np.load(dump + "b.pass1.npy")
                a = (got.view(np.uint32) != ref1.view(np.uint32)).any(-1)
                b = (got.view(np.uint32) != ref1b.view(np.uint32)).any(-1)
                unstable = (ref1.view(np.uint32) != ref1b.view(np.uint32)).any(-1)
                nd = int(((a | b) & ~unstable).sum() + (a & b & unstable).sum())
                res["parity"]["reference_pixels_differing_between_its_own_two_runs"] = int(unstable.sum())
                os.remove(dump + "b.pass1.npy")
        res["parity"].update({"pass1_equals_reference_full_frame": nd == 0, "pass1_pixels_differing": nd})
        if ssaa and os.path.exists(dump + ".frame.npy"):
            ref2 = np.load(dump + ".frame.npy")
            mask = torch.zeros((height, width), dtype=torch.uint8, device="cuda")
            for _ in range(3):      # (whichever way rtx_render_frame settles on: the frames are identical -- and compared)
                gpu_scene.render_frame(fb, mask)
                if gpu_scene.frame_status() != 0:
                    raise RuntimeError("frame kernel gave up")
                got = fb.cpu().numpy()
                nd2 = int((got.view(np.uint32) != ref2.view(np.uint32)).any(-1)[1:, 1:].sum())
                res["parity"]["frame_equals_reference_full_frame"] = bool(res["parity"].get("frame_equals_reference_full_frame", True) and nd2 == 0)
                res["parity"]["frame_pixels_differing"] = max(nd2, res["parity"].get("frame_pixels_differing", 0))
            res["parity"]["ssaa_pixels"] = int(mask.sum())
    except Exception as e:          # noqa: BLE001 -- reported in the line, never hidden
        res["parity"] = {"error": repr(e)}
    finally:
        for suffix in (".pass1.npy", ".frame.npy"):
            try:
                os.remove(dump + suffix)
            except OSError:
                pass
    return res


def cpu_baseline(scene_path, width, height, gpu_scene, target_s=15.0, flags=None):
    """Oracle (CPU restatement: same exhaustive BVH walk, thread per 128x128 tile, all host cores) on a bounded
    sample of the SAME frame: 32-row bands spread evenly over the image, as many as fit ~target_s seconds
    (the whole frame when the host is fast enough).  The rays of the sampled rows are counted by the
    instrumented GPU kernel (tests/test_gpu_parity.py proves those counts identical to the oracle's)."""
    cores = os.cpu_count() or 1
    ref = reference_baseline(scene_path, width, height, gpu_scene, cores, flags=flags) if os.environ.get("BENCH_CPU_BASELINE", "reference") == "reference" else None
    if ref is not None:
        # the oracle port on a shorter sample beside it, so that the line can be compared where oracle/_ref is absent
        ref["port"] = cpu_baseline_port(scene_path, width, height, gpu_scene, target_s=8.0, flags=flags)
        return ref
    return cpu_baseline_port(scene_path, width, height, gpu_scene, target_s, flags=flags)


def cpu_baseline_port(scene_path, width, height, gpu_scene, target_s=15.0, flags=None):
    from oracle import oracle as O
    o = O.OracleScene(scene_path, width, height)
    for k, v in (flags or {}).items():
        O.lib().orc_set_flag(o.h, k.encode(), int(v))
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


# VALU issue peak.  On gfx950 a wave64 VALU instruction does not cost "4 cycles": fp32 add / mul / fma / moves / simple
# integer operations on VGPR sources issue in ~2.4 cycles per SIMD, everything else (any SGPR source, min / max, compares,
# selects, v_readlane, DPP, packed, fp64) in ~4.2, the transcendental unit in ~8.2 (tools/ubench/valu_rate.hip,
# profiles/r02_valu_rate.txt).  The peak of a KERNEL is therefore 1024 SIMDs x 2.4 GHz / the mean issue cost of its
# instructions; that mean is taken from the kernel's ISA (tools/isa_mix.py: static counts by class -- the dynamic mix is
# not observable with the counters at hand), falling back to 4 cycles per instruction when no stamped ISA summary exists.
SIMDS, CLOCK_GHZ = 1024, 2.4
VALU_PEAK_GINSTR = SIMDS * CLOCK_GHZ / 4
PROFILE_TAG = "r06"      # the round whose profiles/ summaries this file quotes (regenerate: tools/pmc_pass1.sh r06, tools/isa_mix.py, tools/issue_account.py)
PMC_JSON = os.path.join(ROOT, "profiles", PROFILE_TAG + "_pass1_pmc.json")
ISA_JSON = os.path.join(ROOT, "profiles", PROFILE_TAG + "_pass1_isa.json")
ACCOUNT_JSON = os.path.join(ROOT, "profiles", PROFILE_TAG + "_issue_account.json")


def stamped(path):
    """A profiles/ summary is quoted only if it was measured on the kernel sources this run uses (tools/srchash.py)."""
    from tools.srchash import source_hash
    try:
        d = json.load(open(path))
    except Exception:
        return None, "missing: regenerate with tools/pmc_pass1.sh / tools/isa_mix.py"
    if d.get("source_hash") != source_hash():
        return None, "stale: measured on sources %s, running %s -- regenerate with tools/pmc_pass1.sh / tools/isa_mix.py" % (d.get("source_hash"), source_hash())
    return d, None


def pmc_roofline(avg_ms, scene_bytes, fb_bytes, kernel="rtxPass1Kernel<false, true, true>", workload="headline"):
    """Hardware-counter side of the roofline of the dominant kernel (rtxPass1Kernel where the frame is three launches,
    rtxFrameKernel where it is one): VALU wave-instructions and HBM-side bytes per launch from profiles/<round>_pass1_pmc.json
    (tools/pmc_pass1.sh: separate rocprofv3 --pmc passes of this command), over the launch duration measured live in THIS run."""
    d, why = stamped(PMC_JSON)
    if d is None:
        return {"counters": None, "note": why}
    # counters belong to a workload: they are quoted for the BASELINE configuration they were collected on, and for the kernel
    # this run actually spent its time in
    w = d.get("workloads", {}).get(workload)
    if w is None:
        return {"counters": None, "note": "%s holds no counters of workload '%s'" % (os.path.basename(PMC_JSON), workload)}
    # the workload runs ONE variant of its dominant kernel (MESH / BOXES template arguments follow the scene): matched by family
    fam = "FrameKernel<" if "FrameKernel" in kernel else ("SsaaKernel<false" if "SsaaKernel" in kernel else ("Sobel" if "Sobel" in kernel else "Pass1Kernel<false"))
    k = [v for n, v in w["kernels"].items() if fam in n]
    kname = [n for n in w["kernels"] if fam in n]
    if not k:
        return {"counters": None, "note": "no %s among the counters of workload '%s' (collected with the frame rendered in %s)" % (kernel, workload, w["frame"])}
    k = k[0]
    valu = k["SQ_INSTS_VALU"]
    # MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports half the bytes of wide coalesced streaming reads (the leaf-reference
    # stream: 16 B per lane) -- doubled; WRITE_SIZE as reported.  KB -> bytes.
    fetch = 2.0 * k["FETCH_SIZE"] * 1024.0
    write = k["WRITE_SIZE"] * 1024.0
    sec = avg_ms * 1e-3
    out = {"valu_instructions": valu, "valu_ginstr_s": valu / sec / 1e9,
           "hbm": {"fetch_bytes": int(fetch), "write_bytes": int(write), "achieved_GBs": round((fetch + write) / sec / 1e9, 1), "peak_GBs": HBM_PEAK_GBS,
                   "frac": round((fetch + write) / sec / 1e9 / HBM_PEAK_GBS, 4),
                   "compulsory_bytes": int(scene_bytes + fb_bytes),
                   "traffic_over_compulsory": round((fetch + write) / max(scene_bytes + fb_bytes, 1), 2),
                   "l2_hit_rate": round(k["TCC_HIT_sum"] / (k["TCC_HIT_sum"] + k["TCC_MISS_sum"]), 3) if "TCC_HIT_sum" in k else None},
           "sq": {c: k[c] for c in k if c.startswith("SQ_")}, "source_hash": d["source_hash"], "kernel": kname[0]}
    # where the wave-cycles go (SQ_WAVE_CYCLES = parked on s_waitcnt + ready but not issued + issuing: MI355X_MICROARCH.md, rocprofv3 PMC slots) and
    # how full the VALU instructions are (SQ_THREAD_CYCLES_VALU / 64 lanes x the quad-cycles VALU instructions were active)
    if all(c in k for c in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY")):
        wc = k["SQ_WAVE_CYCLES"]
        out["wave_cycles"] = {"waiting_for_memory": round(k["SQ_WAIT_ANY"] / wc, 3), "ready_not_issued": round(k["SQ_WAIT_INST_ANY"] / wc, 3),
                              "issuing": round(k["SQ_ACTIVE_INST_ANY"] / wc, 3), "issuing_valu": round(k["SQ_ACTIVE_INST_VALU"] / wc, 3)}
    if "SQ_THREAD_CYCLES_VALU" in k and k.get("SQ_ACTIVE_INST_VALU"):
        out["valu_lane_utilisation"] = round(k["SQ_THREAD_CYCLES_VALU"] / (64.0 * k["SQ_ACTIVE_INST_VALU"]), 3)
    acc, _ = stamped(ACCOUNT_JSON)
    if acc is not None and workload in acc.get("workloads", {}):
        out["useful_valu_frac"] = acc["workloads"][workload].get("useful_valu_frac")
        out["useful_valu_basis"] = acc.get("useful_valu_basis")
    isa, _ = stamped(ISA_JSON)
    if isa is not None:
        out["static_valu_mix"] = isa["valu_fraction_by_class"]
        out["spills"] = {"vgpr": isa.get("vgpr_spill_count"), "sgpr": isa.get("sgpr_spill_count"), "scratch_bytes_per_lane": isa.get("private_segment_fixed_size")}
    return out


CONFIGS = {
    "headline": ("scenes/cfg2_smooth_250k.scene", 4096, 4096),
    "cfg1": ("scenes/cfg1_simple_shapes.scene", 512, 512),
    "cfg2": ("scenes/cfg2_smooth_250k.scene", 1920, 1080),
    "cfg3": ("scenes/cfg3_reflective_refractive.scene", 1920, 1080),
    "cfg4": ("scenes/cfg4_textured_1024.scene", 4096, 4096),
    "cfg5": ("scenes/cfg2_smooth_250k.scene", 8192, 8192),
    "area": ("scenes/area_light.scene", 1920, 1080),      # samples^2 shadow rays per hit (not a BASELINE configuration; VERDICT r4 "missing" item 5)
    # options::useBackfaceCulling = 0 (options.h:27, objects.cpp:75-79): the on / off comparison is one of the four numbers the reference publishes (README.md:56-60)
    # evidence beyond the two synthetic meshes the knobs were fitted on, all in the north-star scene at 4096^2 with the shipped knobs:
    # a second 250k-triangle mesh without pole slivers, and the reference's own models (as the triangles its loader produced: rendering_amd/assets.py)
    "knot": ("scenes/r6_knot_250k.scene", 4096, 4096),
    "ref_bunny": ("scenes/r6_ref_bunny.scene", 4096, 4096), "ref_cow": ("scenes/r6_ref_cow.scene", 4096, 4096),
    "ref_teapot": ("scenes/r6_ref_teapot.scene", 4096, 4096), "ref_sphere": ("scenes/r6_ref_sphere.scene", 4096, 4096),
    "headline_nocull": ("scenes/cfg2_smooth_250k.scene", 4096, 4096),
    "cfg2_nocull": ("scenes/cfg2_smooth_250k.scene", 1920, 1080),
}


def one_launch_hint(scene):
    """True when rtx_render_frame settled on the single launch for this view (nothing to pipeline: the stages already overlap inside it)."""
    try:
        return scene.frame_mode()[0] == 1
    except Exception:
        return False


CONFIG_FLAGS = {"headline_nocull": {"useBackfaceCulling": 0}, "cfg2_nocull": {"useBackfaceCulling": 0}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="headline", choices=sorted(CONFIGS),
                    help="BASELINE.json workload: headline = the metric's own (250k scene at 4096^2, every N); cfg5 = the same scene at 8192^2; cfg1..cfg4 = the other configs")
    ap.add_argument("--scene", default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ssaa", action="store_true")
    ap.add_argument("--verify", dest="verify", action="store_true", default=True,
                    help="N > 1 (default): after the timed region rank 0 also renders the whole frame alone and compares the gathered image with it")
    ap.add_argument("--no-verify", dest="verify", action="store_false")
    ap.add_argument("--pipelined", action="store_true",
                    help="N = 1: after the timed region also measure a SEQUENCE of frames with the SSAA launch of frame k beside pass 1 of frame k + 1 (two streams, two "
                         "framebuffers) -> config.pipelined_ms_per_frame; never `value` (off by default: its overlapping launches would enter a profiler's per-kernel averages)")
    ap.add_argument("--serial-gather", action="store_true",
                    help="N > 1: rtx_gather on the render stream, frame by frame (default: on a second stream, overlapping the next frame)")
    args = ap.parse_args()
    cscene, cw, ch = CONFIGS[args.config]
    custom = bool(args.scene or args.width or args.height or args.no_ssaa)
    args.scene = args.scene or cscene
    args.width = args.width or cw
    args.height = args.height or ch
    if custom and (args.scene, args.width, args.height) != (cscene, cw, ch) or args.no_ssaa:
        args.config = "custom"

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
        # the generated assets the scene names (the two 250k-triangle meshes are made on demand only), plus the small default set
        wanted = [os.path.basename(l.split("=", 1)[1].strip()) for l in open(args.scene) if l.startswith("name=scenes/assets/")]
        assets.ensure()
        assets.ensure([n for n in wanted if n in assets._GENERATORS])
    if world > 1:
        dist.barrier()
    W, H = args.width, args.height
    # scene load = .scene / OBJ / BMP parsing + acceleration-structure build + flatten + upload + view preparation: outside the metric (the
    # reference's "Render scene" timer starts after its loader, scene.cpp:472), reported in config because the reference renders ONE
    # frame per process (main.cpp:15) and the load dwarfs a 4-ms frame
    torch.cuda.synchronize()
    t_load = time.perf_counter()
    flags = CONFIG_FLAGS.get(args.config, {})

    def load_scene():
        sc = RA.Scene(args.scene, W, H, device=local)
        for k, v in flags.items():
            sc.set_flag(k, v)
        sc.gpu()
        return sc
    scene = load_scene()
    torch.cuda.synchronize()
    scene_create_ms = (time.perf_counter() - t_load) * 1e3
    bvh_ms = []
    for oi in range(16):
        try:
            bi = scene.bvh_build_info(oi)
        except Exception:
            bi = None
        if bi is not None and bi[1] >= 0:
            bvh_ms.append({"object": oi, "on_device": bi[0], "ms": round(bi[1], 3)})
    fb = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    mask = torch.zeros((H, W), dtype=torch.uint8, device="cuda")
    ssaa = not args.no_ssaa

    img = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda") if world > 1 else None
    # N > 1: the frame is collected by rtx_gather of the C ABI (RCCL communicator created here, its id broadcast over the
    # torch.distributed group).  BENCH_GATHER=torch selects the same exchange written with torch.distributed
    # point-to-point operations; it is also what runs if the RCCL communicator cannot be created (reported in the line).
    comm, gather_via = None, None
    if world > 1 and backend == "nccl":
        gather_via = "torch.distributed P2P over RCCL"
        if os.environ.get("BENCH_GATHER", "rtx") == "rtx":
            try:
                comm = parallel.make_comm(world, rank, local)
                gather_via = "rtx_gather (C ABI, grouped ncclSend/ncclRecv over RCCL)"
            except Exception as e:          # noqa: BLE001 -- reported, not hidden
                gather_via = "torch.distributed P2P over RCCL (rtx_comm_create failed: %s)" % e
                print("bench.py: " + gather_via, file=sys.stderr, flush=True)
        ok = torch.tensor([1 if comm is not None else 0], dtype=torch.int64, device="cuda")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok[0]) == 0 and comm is not None:      # some rank failed: every rank uses the torch path
            comm.close(); comm = None
            gather_via = "torch.distributed P2P over RCCL (rtx_comm_create failed on another rank)"
    elif world > 1:
        gather_via = "torch.distributed P2P over %s, staged through host memory (functional check)" % backend

    pipe = parallel.FrameGather(scene, comm, img) if (comm is not None and not args.serial_gather) else None
    if pipe is not None:
        gather_via += ", overlapped with the next frame's render (second stream)"

    def step():
        # (fb was zeroed when it was allocated and only ever holds frames of this view and row ownership: the reference's
        # zero-initialised allocation, scene.cpp:599, is outside its timers too)
        parallel.shard_frame(scene, fb, mask, world, rank, ssaa=ssaa, clear=False)
        if world > 1:
            # the frame has to end up in ONE place: quantise to the BGR8 image saveImage writes (4x fewer bytes than
            # the fp32 framebuffer) and send every owned band straight into rank 0's image
            if pipe is not None:
                pipe.submit(fb)      # quantise on this stream, rtx_gather on a second one: it overlaps the next frame's render
                return
            scene.quantize(fb, img)
            if comm is not None:
                comm.gather(scene, img, bottom_up=True)
            elif backend == "nccl":
                parallel.gather_frame(img, world, rank, bottom_up=True)
            else:
                host = img.cpu()
                parallel.gather_frame(host, world, rank, bottom_up=True)
                img.copy_(host)

    def sync():
        if pipe is not None:
            pipe.wait()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # rays / box tests / triangle tests of one frame under reference semantics (instrumented variant, untimed).
    # Pass 1 is counted over the OWNED rows only (halo off), so that the sum over the ranks is the ray count of the
    # frame itself -- the rays of the recomputed halo rows are extra work of the sharding, not units of the metric.
    scene.counters_enable(True)
    scene.set_row_ownership(parallel.band_height(H, world) if world > 1 else 0, world, rank, halo=False)
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
    tot_p1, tot_p2 = int(c1[0]), int(c2[0])      # rays of this rank's pass 1 / SSAA pass
    tot = torch.tensor([int(x) for x in (c1 + c2)] + [moot], dtype=torch.int64, device=rdev)
    if world > 1:
        dist.all_reduce(tot)
    rays_per_frame = int(tot[0])

    # rtx_render_frame measures, on the first warm frames of a view, whether one launch or three are faster for it, from
    # events it reads back without waiting; a few synchronised frames let it settle before the untimed warmup
    for _ in range(6):
        step()
        sync()
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
    # HIP-event durations of every launch of the timed region (on the launch stream: rtx_kernel_time_stats), read before anything else renders
    n1, ms1 = scene.kernel_time_stats(0)
    nS, msS = scene.kernel_time_stats(1)
    n2, ms2 = scene.kernel_time_stats(2)
    n4, ms4 = scene.kernel_time_stats(4)
    frame_status = 0
    if ssaa:
        # 0, or error | 0x100 when the single launch of some timed frame gave up (the last one was rendered again in three launches, earlier
        # ones stayed incomplete: rtx_frame_status raises for those): either way the timed frames were not all whole frames -- no number
        frame_status = scene.frame_status()
        if frame_status != 0:
            raise SystemExit("bench.py: rtx_frame_status = 0x%x -- a timed frame's single launch gave up; the timing is void" % frame_status)

    # N > 1: what the timed region measured is THROUGHPUT when the exchange of frame k runs beside the rendering of frame k + 1 (the default); the LATENCY of one
    # frame -- render this rank's rows, quantise, gather on the render stream, wait -- is measured here, frame by frame (max over the ranks)
    frame_latency_ms = None
    gather_ms_rank = 0.0
    if world > 1:
        def one_frame_serial():
            parallel.shard_frame(scene, fb, mask, world, rank, ssaa=ssaa, clear=False)
            scene.quantize(fb, img)
            if comm is not None:
                comm.gather(scene, img, bottom_up=True)
            elif backend == "nccl":
                parallel.gather_frame(img, world, rank, bottom_up=True)
            else:
                host = img.cpu()
                parallel.gather_frame(host, world, rank, bottom_up=True)
                img.copy_(host)
        sync()
        one_frame_serial(); sync()
        # (the exchange alone: events on the render stream around quantise + gather of every serial frame)
        g0 = [torch.cuda.Event(enable_timing=True) for _ in range(5)]; g1 = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        tl = time.perf_counter()
        for k in range(5):
            parallel.shard_frame(scene, fb, mask, world, rank, ssaa=ssaa, clear=False)
            g0[k].record()
            scene.quantize(fb, img)
            if comm is not None:
                comm.gather(scene, img, bottom_up=True)
            elif backend == "nccl":
                parallel.gather_frame(img, world, rank, bottom_up=True)
            else:
                host = img.cpu()
                parallel.gather_frame(host, world, rank, bottom_up=True)
                img.copy_(host)
            g1[k].record()
            sync()
        lat = torch.tensor([(time.perf_counter() - tl) / 5], dtype=torch.float64, device=rdev)
        dist.all_reduce(lat, op=dist.ReduceOp.MAX)
        frame_latency_ms = float(lat[0]) * 1e3
        gather_ms_rank = sum(a.elapsed_time(b) for a, b in zip(g0, g1)) / 5

    # N = 1, frames in a SEQUENCE: the SSAA launch of frame k (a few thousand latency-bound items: 0.3 of the issue peak) beside pass 1 + Sobel of frame k + 1 on a second
    # stream, two framebuffers -- what a caller rendering a camera path would do.  Reported beside `ms_per_step`, which stays the serial frame (one stream, one
    # framebuffer: every frame complete before the next begins); both framebuffers are compared with the serial frame afterwards.
    pipelined_ms = pipelined_same = None
    if args.pipelined and world == 1 and ssaa and not one_launch_hint(scene):
        fbs = [fb, torch.zeros_like(fb)]; masks = [mask, torch.zeros_like(mask)]
        ref_fb, ref_mask = fb.clone(), mask.clone()
        sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
        free = [None, None]

        def piped(k):
            i = k & 1
            if free[i] is not None:
                sA.wait_event(free[i])          # frame k - 2 has left this framebuffer
            scene.render_pass1(fbs[i], stream=sA)
            scene.sobel(fbs[i], masks[i], stream=sA)
            e = torch.cuda.Event(); e.record(sA)
            sB.wait_event(e)
            scene.render_ssaa(masks[i], fbs[i], stream=sB)
            free[i] = torch.cuda.Event(); free[i].record(sB)
        torch.cuda.synchronize()
        for k in range(4):
            piped(k)
        torch.cuda.synchronize()
        tp = time.perf_counter()
        for k in range(args.steps):
            piped(k)
        torch.cuda.synchronize()
        pipelined_ms = (time.perf_counter() - tp) / args.steps * 1e3
        pipelined_same = bool((fbs[0] == ref_fb).all()) and bool((fbs[1] == ref_fb).all()) and bool((masks[0] == ref_mask).all()) and bool((masks[1] == ref_mask).all())
        del fbs, masks, ref_fb, ref_mask

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
    frame_mode, split_ms, fused_ms = scene.frame_mode() if ssaa else (0, -1.0, -1.0)
    # the dominant kernel = the launch with the largest SUMMED duration over the timed region (on cfg3 that is the SSAA launch,
    # not pass 1): pass 1, Sobel, SSAA, or the single kernel of the frame where that is what ran
    one_launch = ssaa and n4 > n1
    stages = {"rtxPass1Kernel<false, true, true>": (n1, ms1), "rtxSobelKernel": (nS, msS), "rtxSsaaKernel<false, true, true>": (n2, ms2), "rtxFrameKernel<true, true>": (n4, ms4)}
    dom_kernel = max(stages, key=lambda k: stages[k][1])
    n_dom, ms_dom = stages[dom_kernel]
    avg_ms = ms_dom / max(n_dom, 1)
    # N > 1: what the RCCL communicator itself says it is, and the slowest rank's stage times (the frame ends when the slowest rank does; the first real
    # SCALE line should explain itself: VERDICT r5 next 9)
    rccl_ranks = per_rank_max = None
    if world > 1:
        if comm is not None:
            rccl_ranks = comm.info()[0]
        mine = [ms1 / n1 if n1 else 0.0, msS / nS if nS else 0.0, ms2 / n2 if n2 else 0.0, ms4 / n4 if n4 else 0.0, gather_ms_rank]
        mx = torch.tensor(mine, dtype=torch.float64, device=rdev); mn = mx.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX); dist.all_reduce(mn, op=dist.ReduceOp.MIN)
        names = ("pass1_ms", "sobel_ms", "ssaa_ms", "frame_kernel_ms", "quantize_gather_ms")
        per_rank_max = {k: round(float(v), 3) for k, v in zip(names, mx)}
        per_rank_max.update({"min_" + k: round(float(v), 3) for k, v in zip(names, mn)})
    # cold frame: a freshly created scene in the warm process -- its first pass 1 has no tile costs of a previous launch to
    # order its queues by (the reference's use case is one frame per process)
    # The GPU has been idle while the host loaded that scene, and its clocks need a few milliseconds to come back: a second
    # fresh scene is timed (HIP events) directly behind three frames of the warm one -- what the first frame costs in software
    # (no measured tile costs: the estimate of rtx_scene_create orders and splits it; tools/cold_probe.py).
    cold_ms = cold_busy_ms = set_view_ms = new_view_host_ms = new_view_frame_ms = new_view_device_ms = None
    if world == 1:
        scene2 = load_scene()
        torch.cuda.synchronize()
        tc = time.perf_counter()
        parallel.shard_frame(scene2, fb, mask, 1, 0, ssaa=ssaa)
        torch.cuda.synchronize()
        cold_ms = (time.perf_counter() - tc) * 1e3
        scene2.close()
        scene3 = load_scene()
        torch.cuda.synchronize()
        fb3 = torch.zeros_like(fb); mask3 = torch.zeros_like(mask)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            step()
        e0.record()
        parallel.shard_frame(scene3, fb3, mask3, 1, 0, ssaa=ssaa, clear=False)
        e1.record()
        torch.cuda.synchronize()
        cold_busy_ms = e0.elapsed_time(e1)
        # what a NEW VIEW of a loaded scene costs before its first frame (rtx_scene_set_view runs the first-frame cost estimate, builds the
        # tile lists and the camera's copy of the prune records and synchronises the device): wall clock of re-applying the view, outside every frame time above
        torch.cuda.synchronize()
        tv = time.perf_counter()
        scene3.resize(W, H)
        scene3.gpu()
        torch.cuda.synchronize()
        set_view_ms = (time.perf_counter() - tv) * 1e3
        # a NEW view (the camera moved a little): what rtx_scene_set_view costs the HOST (it queues the camera's source records, the cost estimate and the
        # tile lists on the device and returns: round 5), and the first frame of that view on the device (HIP events, directly behind warm frames)
        pos, rot = scene3.camera_pose()
        for _ in range(3):
            step()
        ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        ea.record()
        tv = time.perf_counter()
        scene3.set_camera(pos + np.float32([0.02, 0.01, 0.0]), rot + np.float32([0.0, 0.7, 0.0]))
        scene3.gpu()
        new_view_host_ms = (time.perf_counter() - tv) * 1e3
        eb.record()      # (what the view's preparation costs the DEVICE: source records of the camera, cost estimate, tile lists -- queued, ahead of the first frame)
        e0.record()
        parallel.shard_frame(scene3, fb3, mask3, 1, 0, ssaa=ssaa, clear=False)
        e1.record()
        torch.cuda.synchronize()
        new_view_frame_ms = e0.elapsed_time(e1)
        new_view_device_ms = ea.elapsed_time(eb)
        scene3.close()
        del fb3, mask3
    # The SURVEY 8d byte model (32 B per box test + 40 B per triangle test counted under REFERENCE traversal semantics
    # + 12 B per rendered pixel) is kept as a description of the reference's work -- the kernel shares every fetch among
    # 64 rays and rejects whole groups of triangles without touching them per ray, so it is not what bounds the kernel.
    rendered_px = (W - 1) * (H - 1) / world
    alg_bytes = 32.0 * float(c1[1]) + 40.0 * float(c1[2]) + 12.0 * rendered_px
    walked = rays_per_frame - int(tot[3])
    roof = {"bound": "valu_issue", "kernel": dom_kernel, "avg_launch_ms": round(avg_ms, 3), "launches_timed": n_dom,
            "launch_ms_by_stage": {k: round(v[1] / v[0], 3) for k, v in stages.items() if v[0]},
            "unit": "G wave-instructions/s", "peak": SIMDS * CLOCK_GHZ / 2, "achieved": None, "frac": None, "traffic": None,
            "algorithmic_ref_semantics_bytes": int(alg_bytes), "box_tests": int(c1[1]), "tri_tests": int(c1[2])}
    if world == 1:
        pm = pmc_roofline(avg_ms, scene.scene_bytes(), 12.0 * rendered_px, dom_kernel, args.config)
        # peak = the guide's fp32 issue peak (MI355X_MICROARCH.md: a wave64 v_fma_f32 issues in 2 cycles on a SIMD-32:
        # 1024 SIMDs x 2.4 GHz / 2 = 1228.8 G wave-instructions/s) -- a hardware ceiling; frac = the share of its instruction
        # slots that carry an instruction of this kernel.  Beside it, `mix_weighted`: the same rate against a peak weighted with
        # the kernel's own (static) instruction mix and the issue costs measured by tools/ubench/valu_rate.hip -- a cost model,
        # not a ceiling (it comes out slightly above 1: the model overprices some instruction classes).
        roof["peak"] = SIMDS * CLOCK_GHZ / 2
        roof["peak_basis"] = "1024 SIMDs x 2.4 GHz / 2 cycles per wave64 fp32 VALU instruction (MI355X_MICROARCH.md)"
        isa, _ = stamped(ISA_JSON)
        mean_cycles = (isa or {}).get("kernels", {}).get(dom_kernel, {}).get("valu_issue_cycles_static_mean")
        if pm.get("valu_instructions"):
            roof["achieved"] = round(pm["valu_ginstr_s"], 1)
            roof["frac"] = round(pm["valu_ginstr_s"] / roof["peak"], 4)
            roof["frac_vs_fp32_issue_peak"] = roof["frac"]
            roof["traffic"] = pm["hbm"]["fetch_bytes"] + pm["hbm"]["write_bytes"]
            roof["kernel"] = pm.get("kernel", dom_kernel)
            # SURVEY 8d's reading, spelled out: the HBM roofline of this launch -- measured HBM bytes over the 8 TB/s peak -- and what the survey's
            # byte model of the REFERENCE's traversal (32 B per box test + 40 B per triangle test + 12 B per pixel) would need at this speed, as a
            # multiple of the peak: far above 1 means the kernel does not do the reference's per-ray work (it shares every fetch among 64 rays and
            # rejects whole groups of triangles), which is why the line's `frac` is the VALU-issue fraction.
            # what measures headroom: the share of the issue peak that carries USEFUL arithmetic (box / triangle / shading arithmetic of the reference's own
            # algorithm; the rest is bundle, prune, queue and bookkeeping overhead -- profiles/<round>_issue_account.txt)
            roof["useful_frac"] = round(roof["frac"] * pm["useful_valu_frac"], 4) if pm.get("useful_valu_frac") else None
            roof["hbm_frac"] = pm["hbm"]["frac"]
            roof["ref_semantics_bytes_over_peak"] = round(alg_bytes / (avg_ms * 1e-3) / (HBM_PEAK_GBS * 1e9), 2)
            if mean_cycles:
                mw = SIMDS * CLOCK_GHZ / mean_cycles
                roof["mix_weighted"] = {"peak": round(mw, 1), "frac_unclamped": round(pm["valu_ginstr_s"] / mw, 4),
                                        "basis": "1024 SIMDs x 2.4 GHz / %.2f cycles per VALU instruction (static mix of %s by issue class: %s)" % (
                                            mean_cycles, dom_kernel, isa["kernels"][dom_kernel]["valu_by_issue_cycles"])}
        roof["counters"] = pm
    out = {
        "metric": "Mrays/s + ms/frame at 4096^2, 250k-tri BVH scene",
        # `value` counts the rays the kernels WALK.  The reference also casts `moot_shadow_rays` (stats::raysCasted counts them): shadow rays whose answer cannot
        # change the pixel, which the kernels provably need not trace -- `value_counted` includes them (the reference's own definition of a ray)
        "value": round(walked * args.steps / dt / 1e6, 3),
        "value_counted": round(rays_per_frame * args.steps / dt / 1e6, 3),
        "unit": "Mrays/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s @%dx%d, pass 1%s" % (os.path.basename(args.scene), W, H, " + Sobel-adaptive SSAA" if ssaa else ""),
                   "name": args.config, "flags": flags or None,
                   "rays_per_frame": rays_per_frame, "moot_shadow_rays": int(tot[3]),
                   "walked_rays_per_frame": walked, "counted_mrays_s": round(rays_per_frame * args.steps / dt / 1e6, 3),
                   "parallelism": "rows in %d-row bands over %d GPU(s)%s" % (parallel.band_height(H, world), world, ", BGR8 bands collected on rank 0" if world > 1 else ""),
                   "gather": gather_via, "rccl_ranks": rccl_ranks, "slowest_rank": per_rank_max,
                   # N > 1: ms_per_step is pipelined throughput when the gather overlaps the next frame's render; frame_latency_ms is one frame end to end
                   "ms_per_step_is": None if world == 1 else ("pipelined throughput (gather of frame k beside the render of frame k + 1)" if pipe is not None else "serial frames (render, gather, next frame)"),
                   "frame_latency_ms": None if frame_latency_ms is None else round(frame_latency_ms, 3),
                   # (N = 1) a sequence of frames with the SSAA launch of frame k beside pass 1 of frame k + 1: two streams, two framebuffers; never `value`
                   "pipelined_ms_per_frame": None if pipelined_ms is None else round(pipelined_ms, 3), "pipelined_frames_equal_serial_frame": pipelined_same,
                   "frame": ("one launch (rtxFrameKernel)" if one_launch else "three launches (pass 1, Sobel, SSAA)") if ssaa else "pass 1 only",
                   "measured_three_launches_ms": None if split_ms < 0 else round(split_ms, 3), "measured_one_launch_ms": None if fused_ms < 0 else round(fused_ms, 3),
                   "pass1_ms": round(ms1 / n1, 3) if n1 else None, "ssaa_ms": round(ms2 / n2, 3) if n2 else None,
                   "sobel_ms": round(msS / nS, 3) if nS else None,
                   "pass1_grays_s": round(float(tot_p1) / (ms1 / n1) / 1e6, 2) if n1 else None, "ssaa_grays_s": round(float(tot_p2) / (ms2 / n2) / 1e6, 2) if n2 and ssaa else None,
                   "scene_create_ms": round(scene_create_ms, 1), "bvh_build": bvh_ms,
                   "frame_kernel_ms": round(ms4 / n4, 3) if n4 else None,
                   "cold_frame_ms": None if cold_ms is None else round(cold_ms, 3),
                   "cold_frame_gpu_busy_before_ms": None if cold_busy_ms is None else round(cold_busy_ms, 3),
                   "set_view_ms": None if set_view_ms is None else round(set_view_ms, 3),
                   "new_view_host_ms": None if new_view_host_ms is None else round(new_view_host_ms, 3),
                   "new_view_device_ms": None if new_view_device_ms is None else round(new_view_device_ms, 3),
                   "new_view_first_frame_ms": None if new_view_frame_ms is None else round(new_view_frame_ms, 3),
                   # the reference renders ONE frame per process (main.cpp:15): the first frame of a scene as a rate, beside `value` (the steady state of a view)
                   "cold_frame_mrays_s": None if cold_busy_ms is None else round(walked / cold_busy_ms / 1e3, 1),
                   "cold_frame_wall_mrays_s": None if cold_ms is None else round(walked / cold_ms / 1e3, 1),
                   "pass1_rays_rank0": int(c1[0]), "ssaa_rays_rank0": int(c2[0]), "ssaa_pixels_rank0": int(mask.sum()),
                   "ssaa_box_tests": int(c2[1]), "ssaa_tri_tests": int(c2[2])},
        "roofline": roof,
    }
    if verified is not None:
        out["config"]["gathered_image_equals_single_gpu_image"] = verified
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.scene, W, H, scene, flags=flags)
        par = (out["cpu_baseline"] or {}).get("parity")
        if par:      # the reference's own framebuffers of this very frame against the product path's (reference_baseline)
            for k in ("pass1_equals_reference_full_frame", "frame_equals_reference_full_frame"):
                if k in par:
                    out["config"][k] = par[k]
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
