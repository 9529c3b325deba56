#!/bin/bash
export RTX_ALLOW_ENV_KNOBS=1      # the product ignores RTX_* environment knobs without it (rtx_api.hip readKnobs)
# GPU box: regenerates profiles/<round>_pass1_pmc.json -- hardware counters of the ray kernels of the CURRENT sources,
# stamped with the source hash (tools/srchash.py) so that bench.py only quotes them for the kernels they were measured on.
# One entry per workload of bench.py (--config): the headline, cfg1 ... cfg5 and the area-light scene at 1920x1080 (round 5: every BASELINE
# configuration has a stamped roofline; rounds 2-4 had headline + cfg2 only).  The way the frame is rendered is fixed (RTX_FRAME_MODE) to what
# rtx_render_frame settles on without a profiler attached: one un-profiled bench run per workload asks it first.
# Per workload five rocprofv3 --pmc passes (counters only, no trace domains: gpurun refuses the combination) of
#     python bench.py --config <cfg> --steps 2 --warmup 1 --no-cpu-baseline
# FETCH_SIZE and WRITE_SIZE cannot share a pass (TCC slots); the SQ counters fill two passes (the split of the wave-cycles --
# SQ_WAIT_ANY = parked on s_waitcnt, SQ_WAIT_INST_ANY = ready but not issued, SQ_ACTIVE_INST_ANY = issuing -- and SQ_THREAD_CYCLES_VALU).
# Usage: tools/pmc_pass1.sh <round tag, e.g. r05> [workloads, default: all]
TAG=${1:-r06}
WLS=${2:-"headline cfg1 cfg2 cfg3 cfg4 cfg5 area"}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in $WLS; do
  mode=$(python $R/bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "import json,sys; print('fused' if json.loads(sys.stdin.read())['config']['frame'].startswith('one') else 'split')")
  echo "$cfg $mode" >> $OUT/modes.txt
  for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "sq:SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "sq2:SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_WR" "tcc:TCC_HIT_sum TCC_MISS_sum"; do
    name=${pass%%:*}; cnt=${pass#*:}
    RTX_FRAME_MODE=$mode rocprofv3 --pmc $cnt --output-format csv -d $OUT/$cfg/$name -o $name -- python $R/bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline > $OUT/$cfg.$name.log 2>&1
  done
done
cd $R
python - "$OUT" "$TAG" <<'PY'
import csv, glob, json, os, sys, collections
sys.path.insert(0, "tools")
from srchash import source_hash
out, tag = sys.argv[1], sys.argv[2]
modes = dict(l.split() for l in open(out + "/modes.txt"))
res = {"source_hash": source_hash(),
       "command": "RTX_ALLOW_ENV_KNOBS=1 RTX_FRAME_MODE=<split|fused> rocprofv3 --pmc <counters> -- python bench.py --config <workload> --steps 2 --warmup 1 --no-cpu-baseline",
       "units": "per launch (average over the dispatches of the kernel); FETCH_SIZE / WRITE_SIZE in KB as reported; SQ_*_CYCLES in quad-cycles",
       "workloads": {}}
for cfg, m in modes.items():
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
    for f in glob.glob(out + "/" + cfg + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0].replace("void ", "")
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
            disp[(k, row["Counter_Name"])].add(row["Dispatch_Id"])
    ks = {}
    for k in agg:
        # the product variants of the ray kernels (first template argument STATS = false) and the Sobel kernel
        if "Pass1Kernel<false" in k or "SsaaKernel<false" in k or "FrameKernel<" in k or "Sobel" in k:
            ks[k] = {c: v / max(len(disp[(k, c)]), 1) for c, v in agg[k].items()}
            ks[k]["dispatches"] = max(len(disp[(k, c)]) for c in agg[k])
    res["workloads"][cfg] = {"frame": "one launch" if m == "fused" else "three launches", "kernels": ks}
if "headline" in res["workloads"]:
    res["kernels"] = res["workloads"]["headline"]["kernels"]      # (the headline's, under the key earlier rounds used)
json.dump(res, open("profiles/%s_pass1_pmc.json" % tag, "w"), indent=1)
json.dump(res, open(out + "/%s_pass1_pmc.json" % tag, "w"), indent=1)     # gpurun merges gpurun_out/ back: copy this one to profiles/ and commit it
for cfg, w in res["workloads"].items():
    for k, v in w["kernels"].items():
        if "Sobel" not in k: print(cfg, w["frame"], k, "VALU %.4g" % v.get("SQ_INSTS_VALU", 0), "fetch KB %.4g write KB %.4g" % (v.get("FETCH_SIZE", 0), v.get("WRITE_SIZE", 0)))
PY
