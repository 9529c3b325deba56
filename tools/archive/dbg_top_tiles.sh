# GPU box: wave-level counters of the slowest pass-1 tiles of a view (RTX_DBG build): tools/dbg_top_tiles.sh scene W H [n]
export RTX_ALLOW_ENV_KNOBS=1      # the product ignores RTX_* environment knobs without it (rtx_api.hip readKnobs)
cd ${GRAFT_REPO_ROOT:-.}
python tools/top_tiles.py $1 $2 $3 ${4:-3} > /tmp/top.txt 2> /tmp/top.err; grep -v amdgpu /tmp/top.err
RTX_DEFS="-DRTX_DBG=1" ./build.sh > gpurun_out/build_dbg.log 2>&1
for t in $(cat /tmp/top.txt); do
  ty=${t#*,}
  echo "== tile $t"
  RTX_DBG_TILE=$t RTX_DEBUG_ITEMS=1 python tools/dbg_rows.py $((ty*8)) $((ty*8+8)) $1 $2 $3 2>&1 | grep "wave-level\|rows\|work items\|walk cycles"
done
./build.sh > /dev/null 2>&1
