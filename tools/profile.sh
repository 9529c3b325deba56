#!/bin/bash
# Usage (on the GPU box, via gpurun): tools/profile.sh <tag> [bench args...]
# Writes the rocprofv3 kernel-trace + stats CSVs of one bench.py run under gpurun_out/prof_<tag>/ .
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o $TAG -- python $R/bench.py --no-cpu-baseline "$@" > $OUT/bench.log 2>&1
grep '^{' $OUT/bench.log > $OUT/bench.json
find $OUT -name '*kernel_stats.csv' -exec cat {} \;
