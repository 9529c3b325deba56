#!/bin/bash
# GPU box: every variant under rendering_amd/_variants on several BASELINE configs
cd ${GRAFT_REPO_ROOT:-.}
cp rendering_amd/librtx_hip.so /tmp/librtx_orig.so
for v in rendering_amd/_variants/librtx_*.so; do
  cp $v rendering_amd/librtx_hip.so
  for c in ${CFGS:-headline cfg1 cfg2 cfg3 cfg4 cfg5}; do
    echo -n "$(basename $v) $c: "
    python bench.py --no-cpu-baseline --config $c --steps 5 --warmup 2 2>&1 | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('Mrays/s', d['value'], 'ms/frame', d['ms_per_step'], c['frame'][:12], 'pass1', c['pass1_ms'], 'ssaa', c['ssaa_ms'], 'frame kernel', c['frame_kernel_ms'], 'measured', c['measured_three_launches_ms'], c['measured_one_launch_ms'])"
  done
done
cp /tmp/librtx_orig.so rendering_amd/librtx_hip.so
