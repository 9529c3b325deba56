#!/bin/bash
export RTX_ALLOW_ENV_KNOBS=1      # the product ignores RTX_* environment knobs without it (rtx_api.hip readKnobs)
# GPU box: RTX_DBG build -> wave-level counters of the product pass 1 at the headline, with and without the source copies of the prune records
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04; mkdir -p $O
RTX_DEFS="-DRTX_DBG=1" ./build.sh > $O/build_dbg.log 2>&1
for v in src nosrc; do
  if [ $v = nosrc ]; then export RTX_NO_SRC=1; else unset RTX_NO_SRC; fi
  echo "== $v"; DBG_PRODUCT=1 RTX_DEBUG_ITEMS=1 python tools/dbg_counts.py "$@" 2>&1 | grep -v amdgpu.ids
done
unset RTX_NO_SRC
./build.sh > /dev/null 2>&1
