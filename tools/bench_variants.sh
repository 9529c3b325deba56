#!/bin/bash
# On the GPU box: benchmarks every rendering_amd/_variants/librtx_*.so (A/B of kernel builds in one gpurun call).
cd ${GRAFT_REPO_ROOT:-.}
cp rendering_amd/librtx_hip.so /tmp/librtx_orig.so
for v in rendering_amd/_variants/librtx_*.so; do
  cp $v rendering_amd/librtx_hip.so
  echo -n "$(basename $v): "
  python bench.py --no-cpu-baseline "$@" 2>&1 | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('Mrays/s', d['value'], 'ms/frame', d['ms_per_step'], 'pass1', d['config']['pass1_ms'], 'ssaa', d['config']['ssaa_ms'])"
done
cp /tmp/librtx_orig.so rendering_amd/librtx_hip.so
