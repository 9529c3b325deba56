#!/bin/bash
export RTX_ALLOW_ENV_KNOBS=1      # the product ignores RTX_* environment knobs without it (rtx_api.hip readKnobs)
# GPU box: every variant on coarse frames of the 250k scene (wide bundles) and the headline
cd ${GRAFT_REPO_ROOT:-.}
cp rendering_amd/librtx_hip.so /tmp/librtx_orig.so
for v in rendering_amd/_variants/librtx_*.so; do
  cp $v rendering_amd/librtx_hip.so
  for f in ${FACTORS:-3}; do
  for a in "512 512" "1920 1080" "4096 4096"; do set -- $a
    echo -n "$(basename $v) fat=$f $1x$2: "
    RTX_FAT_FACTOR=$f python bench.py --scene scenes/cfg2_smooth_250k.scene --width $1 --height $2 --no-cpu-baseline --steps 10 --warmup 3 2>&1 | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms/frame', d['ms_per_step'], 'pass1', d['config']['pass1_ms'], 'ssaa', d['config']['ssaa_ms'])"
  done; done
done
cp /tmp/librtx_orig.so rendering_amd/librtx_hip.so
