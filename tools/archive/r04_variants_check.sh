#!/bin/bash
# GPU box: every variant under rendering_amd/_variants through a parity subset (goldens + oracle parity + margins), then the A/B of tools/bench_ab.sh
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04; mkdir -p $O
cp rendering_amd/librtx_hip.so /tmp/librtx_keep.so
for v in rendering_amd/_variants/librtx_*.so; do
  cp $v rendering_amd/librtx_hip.so
  echo "== $(basename $v): $(timeout 600 python -m pytest ${PARITY_TESTS:-tests/test_gpu_golden.py tests/test_gpu_parity.py tests/test_gpu_margins.py} -x -q 2>&1 | grep -E 'passed|failed|error' | tail -1)"
done 2>&1 | tee $O/variants_parity.txt
cp /tmp/librtx_keep.so rendering_amd/librtx_hip.so
REPS=${REPS:-2} CFGS="${CFGS:-headline cfg2 cfg4}" bash tools/bench_ab.sh 2>&1 | tee $O/ab.txt
