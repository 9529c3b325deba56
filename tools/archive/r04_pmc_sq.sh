#!/bin/bash
# GPU box: where the wave-cycles of the ray kernels go (SQ counters; separate --pmc passes, counters only): tools/r04_pmc_sq.sh <tag>
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/pmcsq_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_list.txt 2>&1
for pass in "a:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" \
            "b:SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVES" \
            "c:SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_FLAT" \
            "d:TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum" "e:GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCC_EA0_RDREQ_sum"; do
  name=${pass%%:*}; cnt=${pass#*:}
  rocprofv3 --pmc $cnt --output-format csv -d $OUT/$name -o $name -- python $R/tools/time_stages.py scenes/cfg2_smooth_250k.scene 4096 4096 4 > $OUT/$name.log 2>&1
done
cd $R
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        if "Pass1" in k or "Ssaa" in k:
            agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in agg:
    print(k)
    for c, v in sorted(agg[k].items()):
        print("   %-32s last dispatch %.4g (of %d)" % (c, v[-1], len(v)))
PY
