#!/bin/bash
export RTX_ALLOW_ENV_KNOBS=1      # the product ignores RTX_* environment knobs without it (rtx_api.hip readKnobs)
# GPU box: the round's profiles for the CURRENT sources -- PMC counters (tools/pmc_pass1.sh), bench lines plain and under
# rocprofv3 --kernel-trace --stats (headline + cfg2), all BASELINE configs, RTX_DBG wave-level counts with and without the
# prune records, shard emulation.  Results under gpurun_out/r04/ (copied to profiles/ by tools/r04_copy.sh).
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04; mkdir -p $O
bash tools/pmc_pass1.sh r04 > $O/pmc.log 2>&1
python bench.py > $O/bench_default.log 2>&1; grep '^{' $O/bench_default.log > $O/r04_bench_default.json
python bench.py --config cfg2 --no-cpu-baseline 2>&1 | grep '^{' > $O/r04_bench_cfg2.json
bash tools/profile.sh r04 --steps 5 --warmup 1 > $O/profile_headline.log 2>&1
cp $(find gpurun_out/prof_r04 -name '*kernel_stats.csv' | head -1) $O/r04_kernel_stats.csv; cp gpurun_out/prof_r04/bench.json $O/r04_bench_under_rocprof.json
bash tools/profile.sh r04cfg2 --config cfg2 --steps 5 --warmup 1 > $O/profile_cfg2.log 2>&1
cp $(find gpurun_out/prof_r04cfg2 -name '*kernel_stats.csv' | head -1) $O/r04_kernel_stats_cfg2.csv; cp gpurun_out/prof_r04cfg2/bench.json $O/r04_bench_cfg2_under_rocprof.json
: > $O/r04_configs.txt
for c in cfg1 cfg2 cfg3 cfg4 cfg5; do
  python bench.py --no-cpu-baseline --config $c 2>/dev/null | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']
print('$c', c['workload'], '|', d['value'], 'Mrays/s', d['ms_per_step'], 'ms/frame |', c['frame'], '| pass1', c['pass1_ms'], 'ssaa', c['ssaa_ms'], 'frame kernel', c['frame_kernel_ms'], '| first frame', c['cold_frame_ms'], 'ms, directly behind warm frames', c.get('cold_frame_gpu_busy_before_ms'), '| rays', c['rays_per_frame'])" >> $O/r04_configs.txt
done
for c in headline cfg4 cfg5; do
  RTX_NO_SRC=1 python bench.py --no-cpu-baseline --config $c 2>/dev/null | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']
print('$c WITHOUT the source copies of the prune records (RTX_NO_SRC=1):', d['value'], 'Mrays/s', d['ms_per_step'], 'ms/frame | pass1', c['pass1_ms'], 'ssaa', c['ssaa_ms'])" >> $O/r04_configs.txt
done
(python tools/shard_time.py 2 4 8; python tools/shard_time.py 2 4 8 --size 8192) 2>&1 | grep -v amdgpu > $O/r04_shard_emulation.txt
bash tools/r04_dbg.sh > $O/r04_dbg_counts.txt 2>&1
python tools/cold_probe.py 2>&1 | grep pass1 > $O/r04_cold_probe.txt
python tools/overhead_probe.py 2>&1 | grep -E "pass1|rays" > $O/r04_overhead_probe.txt
python tools/cost_fit.py 2>&1 | grep -E "scene|together" > $O/r04_cost_fit.txt
python -c "
import json
for f in ('r04_bench_default','r04_bench_cfg2'):
    b=json.load(open('$O/%s.json'%f)); r=b['roofline']; print(f, b['value'], b['ms_per_step'], r.get('peak'), r.get('achieved'), r.get('frac'), r.get('frac_vs_fp32_issue_peak'))"
