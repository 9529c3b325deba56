#!/bin/bash
export RTX_ALLOW_ENV_KNOBS=1      # the product ignores RTX_* environment knobs without it (rtx_api.hip readKnobs)
cd ${GRAFT_REPO_ROOT:-.}
RTX_DEFS="-DRTX_DBG=1 $DBG_DEFS" ./build.sh > gpurun_out/build_dbg.log 2>&1
for cfg in "scenes/cfg2_smooth_250k.scene 4096 4096" "scenes/cfg2_smooth_250k.scene 1920 1080" "scenes/cfg4_textured_1024.scene 4096 4096"; do
echo "== $cfg"; DBG_PRODUCT=1 RTX_DEBUG_ITEMS=1 python tools/dbg_counts.py $cfg 2>&1 | grep -E "wave-level|node visits with|instrumented|wide walks"
done
echo "== SSAA headline (counts include pass 1)"; RTX_DEBUG_ITEMS=1 python tools/dbg_ssaa.py 2>&1 | grep -E "wave-level|node visits with|wide walks|ssaa"
./build.sh > /dev/null 2>&1
