#!/bin/bash
export RTX_ALLOW_ENV_KNOBS=1      # the product ignores RTX_* environment knobs without it (rtx_api.hip readKnobs)
# GPU box: wave-level counters of ONE pass-1 tile in RTX_DBG builds with the given extra defines: tools/r04_tile.sh tx,ty "defsA" "defsB" ...
cd ${GRAFT_REPO_ROOT:-.}
t=$1; shift
ty=${t#*,}
for defs in "$@"; do
  RTX_DEFS="-DRTX_DBG=1 $defs" ./build.sh > gpurun_out/build_dbg.log 2>&1
  echo "== tile $t, $defs"
  RTX_DBG_TILE=$t RTX_DEBUG_ITEMS=1 python tools/dbg_rows.py $((ty*8)) $((ty*8+8)) ${SCENE:-scenes/cfg2_smooth_250k.scene} ${W:-4096} ${H:-4096} 2>&1 | grep "wave-level\|rows\|work items\|walk cycles"
done
./build.sh > /dev/null 2>&1
