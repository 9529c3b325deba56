#!/bin/bash
export RTX_ALLOW_ENV_KNOBS=1      # the product ignores RTX_* environment knobs without it (rtx_api.hip readKnobs)
# GPU box: the round's profiles for the CURRENT sources, in two stages (bench.py quotes only summaries stamped with the sources' hash, and two of
# them -- the ISA summary and the issue account -- are made in the build container between the stages by tools/r06_copy.sh):
#   tools/r06_collect.sh 1   PMC counters of every bench workload (tools/pmc_pass1.sh), RTX_DBG wave-level counts of the headline (for the issue account)
#   tools/r06_copy.sh 1      (here)  copies them to profiles/, makes profiles/r06_pass1_isa.json and profiles/r06_issue_account.*
#   tools/r06_collect.sh 2   bench lines of every workload WITH the reference's CPU baseline and whole-frame parity (VERDICT r5 missing 5), the headline under
#                            rocprofv3 --kernel-trace --stats, kernel stats per workload, first frames, shard emulation, acceleration-structure build
#   tools/r06_copy.sh 2      (here)  copies those to profiles/
# Results under gpurun_out/r06/.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r06; mkdir -p $O
WL="headline cfg1 cfg2 cfg3 cfg4 cfg5 area knot ref_bunny ref_cow ref_teapot ref_sphere headline_nocull"
if [ "$1" = 1 ]; then
  bash tools/pmc_pass1.sh r06 "$WL" > $O/pmc.log 2>&1
  RTX_DEFS="-DRTX_DBG=1" ./build.sh > $O/build_dbg.log 2>&1
  (DBG_PRODUCT=1 RTX_DEBUG_ITEMS=1 python tools/dbg_counts.py; python tools/dbg_ssaa_product.py) 2>&1 | grep -v amdgpu.ids > $O/r06_dbg_counts.txt
  ./build.sh > /dev/null 2>&1
  tail -3 $O/pmc.log; head -3 $O/r06_dbg_counts.txt
elif [ "$1" = 3 ]; then
  # (only the runs under rocprofv3 and the default line again: after a change of bench.py alone)
  python bench.py > $O/bench_default.log 2>&1; grep '^{' $O/bench_default.log > $O/r06_bench_default.json
  bash tools/profile.sh r06 --steps 20 --warmup 5 > $O/profile_headline.log 2>&1
  cp $(find gpurun_out/prof_r06 -name '*kernel_stats.csv' | head -1) $O/r06_kernel_stats.csv; cp gpurun_out/prof_r06/bench.json $O/r06_bench_under_rocprof.json
  for c in cfg1 cfg2 cfg3 cfg4 cfg5 area knot ref_bunny headline_nocull; do
    bash tools/profile.sh r06$c --config $c --steps 5 --warmup 1 > $O/profile_$c.log 2>&1
    cp $(find gpurun_out/prof_r06$c -name '*kernel_stats.csv' | head -1) $O/r06_kernel_stats_$c.csv
  done
  head -4 $O/r06_kernel_stats.csv; cat $O/r06_bench_default.json | head -c 400
else
  for c in $WL; do python bench.py --config $c --steps 20 --warmup 5 --pipelined 2> $O/bench_$c.err | grep '^{' > $O/r06_bench_$c.json; done
  python bench.py 2>/dev/null | grep '^{' > $O/r06_bench_default.json
  bash tools/profile.sh r06 --steps 5 --warmup 1 > $O/profile_headline.log 2>&1
  cp $(find gpurun_out/prof_r06 -name '*kernel_stats.csv' | head -1) $O/r06_kernel_stats.csv; cp gpurun_out/prof_r06/bench.json $O/r06_bench_under_rocprof.json
  for c in cfg1 cfg2 cfg3 cfg4 cfg5 area knot ref_bunny headline_nocull; do
    bash tools/profile.sh r06$c --config $c --steps 5 --warmup 1 > $O/profile_$c.log 2>&1
    cp $(find gpurun_out/prof_r06$c -name '*kernel_stats.csv' | head -1) $O/r06_kernel_stats_$c.csv
  done
  python tools/r06_configs_txt.py $O > $O/r06_configs.txt
  (python tools/shard_time.py 2 4 8; python tools/shard_time.py 2 4 8 --size 8192) 2>&1 | grep -v amdgpu > $O/r06_shard_emulation.txt
  python tools/new_view_probe.py 2>&1 | grep -v amdgpu > $O/r06_new_view_probe.txt
  python tools/bvh_build_time.py 2>&1 | grep -v amdgpu | tail -3 > $O/r06_bvh_build_time.txt
  BENCH_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 2>/dev/null | grep '^{' > $O/r06_bench_gloo2_functional.json
  cat $O/r06_configs.txt
fi
