#!/bin/bash
# GPU box: sweeps of the run-time knobs (rtx_api.hip, readKnobs) on the shipped library: fat factor of the bundle split, SSAA heavy-tile threshold.
# Usage: tools/knob_sweep.sh [config]
cd ${GRAFT_REPO_ROOT:-.}
CFG=${1:-headline}
export RTX_ALLOW_ENV_KNOBS=1
run() { python bench.py --no-cpu-baseline --config $CFG --steps 10 --warmup 2 2>/dev/null | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']
print('$1', 'ms/frame', d['ms_per_step'], 'pass1', c['pass1_ms'], 'ssaa', c['ssaa_ms'], 'frame kernel', c['frame_kernel_ms'])"; }
python bench.py --no-cpu-baseline --config $CFG --steps 3 --warmup 1 > /dev/null 2>&1
for rep in 1 2; do
run "default           "
for f in 1.5 2 4 6 0; do RTX_FAT_FACTOR=$f run "fat_factor=$f      "; done
for t in 9000 12000 25000 40000; do RTX_SSAA_HEAVY_TICKS=$t run "ssaa_heavy_ticks=$t"; done
done
