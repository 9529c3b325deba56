#!/bin/bash
# GPU box: what the rounds cost outside the walk, with counters -- pass 1 of the headline frame and of the same frame with the mesh moved behind the camera
# (tools/overhead_probe.py), SQ counters of the last launch of each.   tools/r05_overhead_pmc.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r05/overhead_pmc; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT -o q -- python $R/tools/overhead_probe.py > $OUT/log.txt 2>&1
cd $R
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
rows = collections.defaultdict(dict)
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "Pass1Kernel<false" in row["Kernel_Name"]:
            rows[int(row["Dispatch_Id"])][row["Counter_Name"]] = float(row["Counter_Value"])
ids = sorted(rows)
# the product launches of the first scene end where the dispatch ids jump (scene load in between); take the last product launch of each half
half = len(ids) // 2
for name, i in (("real frame", ids[half - 1]), ("mesh behind the camera", ids[-1])):
    g = rows[i]
    print("%-24s VALU %.4g SALU %.4g | wave-cycles %.4g: wait-memory %.1f%% issue-stall %.1f%% active %.1f%% (VALU %.1f%%) | lanes/VALU %.3f" % (
        name, g["SQ_INSTS_VALU"], g["SQ_INSTS_SALU"], g["SQ_WAVE_CYCLES"], 100 * g["SQ_WAIT_ANY"] / g["SQ_WAVE_CYCLES"], 100 * g["SQ_WAIT_INST_ANY"] / g["SQ_WAVE_CYCLES"],
        100 * g["SQ_ACTIVE_INST_ANY"] / g["SQ_WAVE_CYCLES"], 100 * g["SQ_ACTIVE_INST_VALU"] / g["SQ_WAVE_CYCLES"], g["SQ_THREAD_CYCLES_VALU"] / g["SQ_INSTS_VALU"] / 64))
print(open(out + "/log.txt").read()[-600:])
PY
