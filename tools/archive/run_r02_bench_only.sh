#!/bin/bash
# GPU box: only the bench lines of the round's profiles (after tools/isa_mix.py and the PMC file are in place for the
# current sources): default, cfg2, and both under rocprofv3 --kernel-trace --stats.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r02; mkdir -p $O
python bench.py > $O/bench_default.log 2>&1; grep '^{' $O/bench_default.log > $O/r02_bench_default.json
python bench.py --config cfg2 --no-cpu-baseline 2>&1 | grep '^{' > $O/r02_bench_cfg2.json
bash tools/profile.sh r02 --steps 5 --warmup 1 > $O/profile_headline.log 2>&1
cp $(find gpurun_out/prof_r02 -name '*kernel_stats.csv' | head -1) $O/r02_kernel_stats.csv; cp gpurun_out/prof_r02/bench.json $O/r02_bench_under_rocprof.json
bash tools/profile.sh r02cfg2 --config cfg2 --steps 5 --warmup 1 > $O/profile_cfg2.log 2>&1
cp $(find gpurun_out/prof_r02cfg2 -name '*kernel_stats.csv' | head -1) $O/r02_kernel_stats_cfg2.csv; cp gpurun_out/prof_r02cfg2/bench.json $O/r02_bench_cfg2_under_rocprof.json
python -c "
import json
for f in ('r02_bench_default','r02_bench_cfg2'):
    b=json.load(open('$O/%s.json'%f)); r=b['roofline']; print(f, b['value'], b['ms_per_step'], r['peak'], r['achieved'], r['frac'], r.get('frac_unclamped'))"
