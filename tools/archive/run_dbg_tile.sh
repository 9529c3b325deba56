# GPU box: counters of single pass-1 tiles (RTX_DBG build): tools/run_dbg_tile.sh "tx,ty tx,ty ..."
export RTX_ALLOW_ENV_KNOBS=1      # the product ignores RTX_* environment knobs without it (rtx_api.hip readKnobs)
RTX_DEFS="-DRTX_DBG=1" ./build.sh > gpurun_out/build_dbg.log 2>&1
for t in $1; do
  ty=${t#*,}
  echo "== tile $t"
  RTX_DBG_TILE=$t RTX_DEBUG_ITEMS=1 python tools/dbg_rows.py $((ty*8)) $((ty*8+8)) 2>&1 | grep "wave-level\|rows"
done
./build.sh > /dev/null 2>&1
