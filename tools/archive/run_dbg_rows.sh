# GPU box: RTX_DBG counters of a pole row and a centre row of the headline frame (product variant)
export RTX_ALLOW_ENV_KNOBS=1      # the product ignores RTX_* environment knobs without it (rtx_api.hip readKnobs)
RTX_DEFS="-DRTX_DBG=1" ./build.sh > gpurun_out/build_dbg.log 2>&1
RTX_DEBUG_ITEMS=1 python tools/dbg_rows.py 3168 3176 2>&1 | grep -v amdgpu
RTX_DEBUG_ITEMS=1 python tools/dbg_rows.py 2048 2056 2>&1 | grep -v amdgpu
./build.sh > /dev/null 2>&1
