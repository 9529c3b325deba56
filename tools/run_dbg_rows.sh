RTX_DEFS="-DRTX_DBG=1" ./build.sh > gpurun_out/build_dbg.log 2>&1
RTX_COOP_TICKS=4294967295 RTX_DEBUG_ITEMS=1 python tools/dbg_rows.py 3168 3176 2>&1 | grep -v amdgpu
RTX_COOP_TICKS=4294967295 RTX_DEBUG_ITEMS=1 python tools/dbg_rows.py 2048 2056 2>&1 | grep -v amdgpu
./build.sh > /dev/null 2>&1
