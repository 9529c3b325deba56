#!/bin/bash
# GPU box: A/B of the variants under rendering_amd/_variants: REPS interleaved rounds over CFGS, best ms/frame and best
# pass-1 / frame-kernel time per (variant, config).  One warm-up run first (the box's clocks).
cd ${GRAFT_REPO_ROOT:-.}
cp rendering_amd/librtx_hip.so /tmp/librtx_orig.so
python bench.py --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2>&1
rm -f /tmp/ab.txt
for rep in $(seq 1 ${REPS:-3}); do
for v in rendering_amd/_variants/librtx_*.so; do
  cp $v rendering_amd/librtx_hip.so
  for c in ${CFGS:-headline cfg2}; do
    python bench.py --no-cpu-baseline --config $c --steps 5 --warmup 2 2>&1 | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']
print('$(basename $v)', '$c', d['ms_per_step'], c['pass1_ms'] or c['frame_kernel_ms'], c['ssaa_ms'] or 0, c['frame'][:3])" >> /tmp/ab.txt
  done
done; done
cp /tmp/librtx_orig.so rendering_amd/librtx_hip.so
python - <<'PY'
import collections
r = collections.defaultdict(list)
for ln in open('/tmp/ab.txt'):
    v, c, ms, k, ss, mode = ln.split()
    r[(c, v)].append((float(ms), float(k), float(ss), mode))
for (c, v), xs in sorted(r.items()):
    print("%-9s %-18s ms/frame best %.3f median %.3f | main kernel best %.3f | ssaa best %.3f | %s" % (c, v, min(x[0] for x in xs), sorted(x[0] for x in xs)[len(xs) // 2], min(x[1] for x in xs), min(x[2] for x in xs), ",".join(x[3] for x in xs)))
PY
