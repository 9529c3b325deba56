#!/bin/bash
# GPU box: A/B of one environment knob of the library: tools/r04_env_ab.sh NAME "v1 v2 ..." ["cfg ..."] -- best / median ms per frame and kernel times of REPS interleaved runs
cd ${GRAFT_REPO_ROOT:-.}
NAME=$1; VALS=$2; CFGS=${3:-headline cfg4 cfg5}
python bench.py --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2>&1
rm -f /tmp/envab.txt
for rep in $(seq 1 ${REPS:-2}); do for v in $VALS; do for c in $CFGS; do
  env $NAME=$v python bench.py --no-cpu-baseline --config $c --steps 5 --warmup 2 2>&1 | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']
print('$NAME=$v', '$c', d['ms_per_step'], c['pass1_ms'] or c['frame_kernel_ms'], c['ssaa_ms'] or 0, c['frame'][:3])" >> /tmp/envab.txt
done; done; done
python - <<'PY'
import collections
r = collections.defaultdict(list)
for ln in open('/tmp/envab.txt'):
    v, c, ms, k, ss, mode = ln.split()
    r[(c, v)].append((float(ms), float(k), float(ss), mode))
for (c, v), xs in sorted(r.items()):
    print("%-9s %-22s ms/frame best %.3f median %.3f | main kernel best %.3f | ssaa best %.3f | %s" % (c, v, min(x[0] for x in xs), sorted(x[0] for x in xs)[len(xs) // 2], min(x[1] for x in xs), min(x[2] for x in xs), ",".join(x[3] for x in xs)))
PY
