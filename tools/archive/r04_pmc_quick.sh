#!/bin/bash
# GPU box: SQ_INSTS_VALU / wave-cycle split of pass 1 for the current librtx_hip.so and for every variant: tools/r04_pmc_quick.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
cp rendering_amd/librtx_hip.so /tmp/librtx_orig.so
for v in /tmp/librtx_orig.so rendering_amd/_variants/librtx_*.so; do
  [ -f $v ] || continue
  cp $v rendering_amd/librtx_hip.so 2>/dev/null
  OUT=$R/gpurun_out/pmcq/$(basename $v .so); rm -rf $OUT; mkdir -p $OUT
  (cd /tmp && TMPDIR=/tmp rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT -o q -- python $R/tools/time_stages.py scenes/cfg2_smooth_250k.scene 4096 4096 6 > $OUT/log.txt 2>&1)
  python - "$OUT" "$(basename $v)" <<'PY'
import csv, glob, sys, collections
out, name = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "Pass1" in row["Kernel_Name"]:
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
g = {c: v[-1] for c, v in agg.items()}
t = [l for l in open(out + "/log.txt") if "frames" in l]
print(name, "| VALU %.4g SALU %.4g | wave-cycles %.4g: wait-memory %.1f%% issue-stall %.1f%% active %.1f%% (VALU %.1f%%) | thread-cycles/VALU/64 %.3f |" % (
    g["SQ_INSTS_VALU"], g["SQ_INSTS_SALU"], g["SQ_WAVE_CYCLES"], 100 * g["SQ_WAIT_ANY"] / g["SQ_WAVE_CYCLES"], 100 * g["SQ_WAIT_INST_ANY"] / g["SQ_WAVE_CYCLES"],
    100 * g["SQ_ACTIVE_INST_ANY"] / g["SQ_WAVE_CYCLES"], 100 * g["SQ_ACTIVE_INST_VALU"] / g["SQ_WAVE_CYCLES"], g["SQ_THREAD_CYCLES_VALU"] / g["SQ_INSTS_VALU"] / 64), t[-1].strip() if t else "")
PY
done
cp /tmp/librtx_orig.so rendering_amd/librtx_hip.so
