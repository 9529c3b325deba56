#!/bin/bash
# GPU box: parity suite (log kept) + bench lines of the given configs
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; grep -E "passed|failed|error" $O/gputests.log | tail -3
for c in ${CONFIGS:-headline cfg2 cfg4 cfg5}; do
  python bench.py --no-cpu-baseline --config $c 2>$O/bench_$c.err | grep '^{' > $O/bench_$c.json
  python -c "import json,sys; d=json.load(open('$O/bench_$c.json')); print('$c', 'Mrays/s', d['value'], 'ms/frame', d['ms_per_step'], 'pass1', d['config']['pass1_ms'], 'ssaa', d['config']['ssaa_ms'], 'cold', d['config']['cold_frame_ms'], d['config'].get('frame'))"
done
