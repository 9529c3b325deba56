#!/bin/bash
# GPU box: tools/overhead_probe.py plain (times) and under rocprofv3 --pmc (VALU instructions per launch)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r03; mkdir -p $O
cd $R; python tools/overhead_probe.py 2>&1 | grep -E "pass1|rays"
cd /tmp && export TMPDIR=/tmp
rm -rf $O/ovh
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $O/ovh -o ovh -- python $R/tools/overhead_probe.py > $O/ovh.log 2>&1
cd $R
python - <<PY
import csv, glob
rows = [r for f in glob.glob("gpurun_out/r03/ovh/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f))]
for r in rows:
    if "Pass1Kernel<false, true, true>" in r["Kernel_Name"]:
        print(r["Dispatch_Id"], r["Counter_Name"], r["Counter_Value"])
PY
