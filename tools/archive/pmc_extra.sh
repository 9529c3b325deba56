#!/bin/bash
export RTX_ALLOW_ENV_KNOBS=1      # the product ignores RTX_* environment knobs without it (rtx_api.hip readKnobs)
# GPU box: a look at counters outside the committed set (instruction cache, scalar cache, LDS, waits) for the headline's pass 1.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/pmc_extra
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/avail.txt 2>&1
grep -o "SQC_[A-Z_0-9]*\|SQ_[A-Z_0-9]*" $OUT/avail.txt | sort -u > $OUT/names.txt
i=0
for cnt in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" \
           "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE" \
           "SQ_IFETCH SQ_WAIT_ANY SQ_WAIT_IFETCH SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_ACTIVE_INST_FLAT"; do
  i=$((i+1))
  RTX_FRAME_MODE=split timeout 300 rocprofv3 --pmc $cnt --output-format csv -d $OUT/p$i -o p$i -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/p$i.log 2>&1
done
cd $R
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); disp[(k, row["Counter_Name"])].add(row["Dispatch_Id"])
for k in agg:
    if "Pass1Kernel<false, true, true>" in k or "SsaaKernel<false, true, true>" in k:
        print(k)
        for c, v in sorted(agg[k].items()): print("   %-32s %.4g" % (c, v / max(len(disp[(k, c)]), 1)))
PY
