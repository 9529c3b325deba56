#!/bin/bash
# Usage (GPU box): tools/pmc.sh <tag> "<counters>" [bench args]   -- one rocprofv3 --pmc pass (no trace domains)
TAG=$1; CNT=$2; shift; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CNT --output-format csv -d $OUT -o $TAG -- python $R/bench.py --no-cpu-baseline "$@" > $OUT/bench.log 2>&1
python - <<PY
import csv, glob, collections
f = glob.glob("$OUT/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
seen=set()
for row in csv.DictReader(open(f[0])):
    k = row["Kernel_Name"][:40]
    agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
    key=(k,row["Dispatch_Id"])
    if key not in seen: seen.add(key); n[k]+=1
for k in agg:
    print(k, "dispatches", n[k], {c: "%.4g" % (v / n[k]) for c, v in agg[k].items()})
PY
