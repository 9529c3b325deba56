#!/bin/bash
# GPU box: everything the round's profiles/ are made of, for the CURRENT sources (run last, after the final kernel change):
#   tests, PMC counters (two workloads), rocprofv3 kernel stats of the headline and of cfg2, the default bench line (with the
#   CPU baseline), the bench line of every BASELINE configuration, the N-GPU projections, the BVH build times.
# Results land in gpurun_out/r02/ ; copy them to profiles/ (tools/collect_r02.sh) and commit.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r02; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 | tee $O/tests.txt
bash tools/pmc_pass1.sh r02 > $O/pmc.log 2>&1; cp gpurun_out/pmc_r02/r02_pass1_pmc.json $O/r02_pass1_pmc.json; cp $O/r02_pass1_pmc.json profiles/r02_pass1_pmc.json
python bench.py > $O/bench_default.log 2>&1; grep '^{' $O/bench_default.log > $O/r02_bench_default.json
python bench.py --config cfg2 --no-cpu-baseline 2>&1 | grep '^{' > $O/r02_bench_cfg2.json
bash tools/profile.sh r02 --steps 5 --warmup 1 > $O/profile_headline.log 2>&1
cp $(find gpurun_out/prof_r02 -name '*kernel_stats.csv' | head -1) $O/r02_kernel_stats.csv; cp gpurun_out/prof_r02/bench.json $O/r02_bench_under_rocprof.json
bash tools/profile.sh r02cfg2 --config cfg2 --steps 5 --warmup 1 > $O/profile_cfg2.log 2>&1
cp $(find gpurun_out/prof_r02cfg2 -name '*kernel_stats.csv' | head -1) $O/r02_kernel_stats_cfg2.csv; cp gpurun_out/prof_r02cfg2/bench.json $O/r02_bench_cfg2_under_rocprof.json
bash tools/bench_configs.sh 2>&1 | grep scene > $O/r02_configs.txt
python tools/shard_time.py 2 4 8 2>&1 | grep -v amdgpu > $O/r02_shard_emulation.txt
python tools/shard_time.py 2 4 8 --size 8192 2>&1 | grep -v amdgpu >> $O/r02_shard_emulation.txt
python tools/bvh_build_time.py 2>&1 | grep -v amdgpu > $O/r02_bvh_build_time.txt
python tools/small_frame_probe.py 2>&1 | grep scene > $O/r02_small_frames.txt
python tools/srchash.py > $O/source_hash.txt
tail -n +1 $O/tests.txt $O/r02_configs.txt $O/r02_shard_emulation.txt $O/r02_bvh_build_time.txt $O/source_hash.txt
