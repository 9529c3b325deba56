set -x
python tools/tile_cost.py > gpurun_out/tile_cost.txt 2>&1
RTX_DEFS=-DRTX_DBG=2 ./build.sh > gpurun_out/build_dbg.log 2>&1
RTX_DEBUG_ITEMS=1 python tools/dbg_counts.py > gpurun_out/dbg_counts.txt 2>&1
./build.sh > /dev/null 2>&1
python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_r02_base.json 2>&1
tail -3 gpurun_out/tile_cost.txt; cat gpurun_out/dbg_counts.txt | tail -30; cat gpurun_out/bench_r02_base.json | tail -2
