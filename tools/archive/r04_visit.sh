#!/bin/bash
export RTX_ALLOW_ENV_KNOBS=1      # the product ignores RTX_* environment knobs without it (rtx_api.hip readKnobs)
# GPU box: where the cycles of a node visit go (RTX_DBG): the whole headline pass 1 (5 waves per SIMD) and one tile alone
cd ${GRAFT_REPO_ROOT:-.}
RTX_DEFS="-DRTX_DBG=1 $DBG_DEFS" ./build.sh > gpurun_out/build_dbg.log 2>&1
echo "== whole frame"; DBG_PRODUCT=1 RTX_DEBUG_ITEMS=1 python tools/dbg_counts.py 2>&1 | grep -E "wave-level|node visits|walk cycles|instrumented|wide walks"
for t in ${TILES:-258,389 300,256}; do
  ty=${t#*,}
  echo "== tile $t alone"
  RTX_DBG_TILE=$t RTX_DEBUG_ITEMS=1 python tools/dbg_rows.py $((ty*8)) $((ty*8+8)) 2>&1 | grep -E "wave-level|node visits with|walk cycles|rows|wide walks"
done
./build.sh > /dev/null 2>&1
