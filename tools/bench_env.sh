#!/bin/bash
# GPU box: bench.py under several values of one environment knob: tools/bench_env.sh VAR "v1 v2 ..." [bench args]
VAR=$1; VALS=$2; shift; shift
for v in $VALS; do
  echo -n "$VAR=$v: "
  env $VAR=$v python bench.py --no-cpu-baseline "$@" 2>&1 | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('Mrays/s', d['value'], 'ms/frame', d['ms_per_step'], 'pass1', d['config']['pass1_ms'], 'ssaa', d['config']['ssaa_ms'], 'cold', d['config']['cold_frame_ms'])"
done
