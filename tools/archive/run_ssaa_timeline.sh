#!/bin/bash
export RTX_ALLOW_ENV_KNOBS=1      # the product ignores RTX_* environment knobs without it (rtx_api.hip readKnobs)
# GPU box: RTX_DBG build -> SSAA timeline, then the product build again
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
RTX_DEFS="-DRTX_DBG=1" ./build.sh > gpurun_out/build_dbg.log 2>&1 || { tail -20 gpurun_out/build_dbg.log; exit 1; }
python tools/ssaa_timeline.py "$@" 2>&1 | grep -v amdgpu.ids
./build.sh > /dev/null 2>&1
