#!/bin/bash
export RTX_ALLOW_ENV_KNOBS=1      # the product ignores RTX_* environment knobs without it (rtx_api.hip readKnobs)
# GPU box: parity suite + headline bench with and without the prune records
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; tail -5 $O/gputests.log
for v in prune noprune; do
  if [ $v = noprune ]; then export RTX_NO_PRUNE=1; else unset RTX_NO_PRUNE; fi
  python bench.py --no-cpu-baseline "$@" 2>$O/bench_$v.err | grep '^{' > $O/bench_$v.json
  python -c "import json,sys; d=json.load(open('$O/bench_$v.json')); print('$v', 'Mrays/s', d['value'], 'ms/frame', d['ms_per_step'], 'pass1', d['config']['pass1_ms'], 'ssaa', d['config']['ssaa_ms'], 'cold', d['config']['cold_frame_ms'], d['config'].get('frame'))"
done
