#!/bin/bash
export RTX_ALLOW_ENV_KNOBS=1      # the product ignores RTX_* environment knobs without it (rtx_api.hip readKnobs)
# GPU box: parity suite (quick subset or all) + A/B of rendering_amd/_variants + RTX_DBG counts of the default build
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q ${PYTEST_ARGS:-} > $O/gputest.log 2>&1; echo "pytest rc $?" >> $O/gputest.log
grep -E "passed|failed|rc " $O/gputest.log
REPS=${REPS:-2} CFGS="${CFGS:-headline cfg2 cfg4}" bash tools/bench_ab.sh 2>&1 | tee $O/ab.txt
if [ -n "$DBG" ]; then
RTX_DEFS="-DRTX_DBG=1 $DBG_DEFS" ./build.sh > $O/build_dbg.log 2>&1
DBG_PRODUCT=1 RTX_DEBUG_ITEMS=1 python tools/dbg_counts.py 2>&1 | grep -v amdgpu.ids > $O/dbg_counts.txt
./build.sh > /dev/null 2>&1
grep -E "wave-level|instrumented" $O/dbg_counts.txt
fi
