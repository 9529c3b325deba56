# GPU box: RTX_DBG build -> wave-level counters of one instrumented pass 1, then restores the product build
export RTX_ALLOW_ENV_KNOBS=1      # the product ignores RTX_* environment knobs without it (rtx_api.hip readKnobs)
RTX_DEFS="-DRTX_DBG=${1:-1}" ./build.sh > gpurun_out/build_dbg.log 2>&1
DBG_PRODUCT=$2 RTX_DEBUG_ITEMS=1 python tools/dbg_counts.py 2>&1 | grep -v amdgpu.ids
./build.sh > /dev/null 2>&1
