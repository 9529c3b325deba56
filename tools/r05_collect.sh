#!/bin/bash
export RTX_ALLOW_ENV_KNOBS=1      # the product ignores RTX_* environment knobs without it (rtx_api.hip readKnobs)
# GPU box: the round's profiles for the CURRENT sources -- PMC counters of every workload (tools/pmc_pass1.sh), bench lines plain (all configs, with their stamped
# rooflines) and under rocprofv3 --kernel-trace --stats, RTX_DBG wave-level counts, shard emulation, first frames, acceleration-structure build.
# Results under gpurun_out/r05/ (copied to profiles/ by tools/r05_copy.sh).
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05; mkdir -p $O
bash tools/pmc_pass1.sh r05 > $O/pmc.log 2>&1
python bench.py > $O/bench_default.log 2>&1; grep '^{' $O/bench_default.log > $O/r05_bench_default.json
for c in cfg1 cfg2 cfg3 cfg4 cfg5 area; do python bench.py --config $c --no-cpu-baseline 2>/dev/null | grep '^{' > $O/r05_bench_$c.json; done
bash tools/profile.sh r05 --steps 5 --warmup 1 > $O/profile_headline.log 2>&1
cp $(find gpurun_out/prof_r05 -name '*kernel_stats.csv' | head -1) $O/r05_kernel_stats.csv; cp gpurun_out/prof_r05/bench.json $O/r05_bench_under_rocprof.json
for c in cfg1 cfg2 cfg3 cfg4 cfg5 area; do
  bash tools/profile.sh r05$c --config $c --steps 5 --warmup 1 > $O/profile_$c.log 2>&1
  cp $(find gpurun_out/prof_r05$c -name '*kernel_stats.csv' | head -1) $O/r05_kernel_stats_$c.csv
done
: > $O/r05_configs.txt
for c in headline cfg1 cfg2 cfg3 cfg4 cfg5 area; do
  f=$O/r05_bench_$c.json; [ $c = headline ] && f=$O/r05_bench_default.json
  python -c "
import json,sys; d=json.loads(open('$f').read()); c=d['config']; r=d['roofline']
print('$c', c['workload'], '|', d['value'], 'Mrays/s', d['ms_per_step'], 'ms/frame |', c['frame'], '| pass1', c['pass1_ms'], 'ssaa', c['ssaa_ms'], 'frame kernel', c['frame_kernel_ms'], '| cold scene', c.get('cold_frame_gpu_busy_before_ms'), 'ms behind warm frames,', c['cold_frame_ms'], 'wall; new view: host', c.get('new_view_host_ms'), 'ms, first frame', c.get('new_view_first_frame_ms'), '| rays', c['rays_per_frame'], '| roofline', r.get('kernel'), 'frac', r.get('frac'), 'hbm_frac', r.get('hbm_frac'), 'ref_semantics_bytes_over_peak', r.get('ref_semantics_bytes_over_peak'))" >> $O/r05_configs.txt
done
(python tools/shard_time.py 2 4 8; python tools/shard_time.py 2 4 8 --size 8192) 2>&1 | grep -v amdgpu > $O/r05_shard_emulation.txt
python tools/shard_stages.py 2>&1 | grep -v amdgpu > $O/r05_shard_stages.txt
python tools/new_view_probe.py 2>&1 | grep -v amdgpu > $O/r05_new_view_probe.txt; python tools/new_view_probe.py scenes/cfg2_smooth_250k.scene 8192 8192 2>&1 | grep -v amdgpu >> $O/r05_new_view_probe.txt
python tools/cold_probe.py 2>&1 | grep pass1 > $O/r05_cold_probe.txt
python tools/bvh_build_time.py 2>&1 | grep -v amdgpu | tail -3 > $O/r05_bvh_build_time.txt
python tools/cost_fit.py 2>&1 | grep -E "scene|together" > $O/r05_cost_fit.txt
RTX_DEFS="-DRTX_DBG=1" ./build.sh > $O/build_dbg.log 2>&1
(DBG_PRODUCT=1 RTX_DEBUG_ITEMS=1 python tools/dbg_counts.py; python tools/dbg_ssaa_product.py) 2>&1 | grep -v amdgpu.ids > $O/r05_dbg_counts.txt
./build.sh > /dev/null 2>&1
python -c "
import json
for f in ('r05_bench_default','r05_bench_cfg2'):
    b=json.load(open('$O/%s.json'%f)); r=b['roofline']; print(f, b['value'], b['ms_per_step'], r.get('peak'), r.get('achieved'), r.get('frac'), r.get('hbm_frac'))"
