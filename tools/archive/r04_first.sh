#!/bin/bash
export RTX_ALLOW_ENV_KNOBS=1      # the product ignores RTX_* environment knobs without it (rtx_api.hip readKnobs)
# GPU box: first look at a change -- the GPU suite, the headline bench with and without the source copies, RTX_DBG wave-level counts.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; echo "pytest rc $?" >> $O/gputest.log
tail -5 $O/gputest.log
for v in src nosrc; do
  if [ $v = nosrc ]; then export RTX_NO_SRC=1; else unset RTX_NO_SRC; fi
  python bench.py --no-cpu-baseline 2>/dev/null | grep '^{' > $O/bench_$v.json
  python -c "
import json; d=json.load(open('$O/bench_$v.json')); c=d['config']
print('$v', d['value'], 'Mrays/s', d['ms_per_step'], 'ms/frame |', c['frame'], '| pass1', c['pass1_ms'], 'ssaa', c['ssaa_ms'])"
done
unset RTX_NO_SRC
RTX_DEFS="-DRTX_DBG=1" ./build.sh > $O/build_dbg.log 2>&1
for v in src nosrc; do
  if [ $v = nosrc ]; then export RTX_NO_SRC=1; else unset RTX_NO_SRC; fi
  echo "== $v"; DBG_PRODUCT=1 RTX_DEBUG_ITEMS=1 python tools/dbg_counts.py 2>&1 | grep -v amdgpu.ids
done > $O/dbg_counts.txt 2>&1
unset RTX_NO_SRC
./build.sh > /dev/null 2>&1
grep -E "wave-level|==|instrumented" $O/dbg_counts.txt
