#!/bin/bash
export RTX_ALLOW_ENV_KNOBS=1      # the product ignores RTX_* environment knobs without it (rtx_api.hip readKnobs)
# GPU box: SSAA stage with 4 / 2 / 1 pixels per wave on the slow tiles, whole frame and one-eighth parts
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04; mkdir -p $O
cp rendering_amd/librtx_hip.so /tmp/librtx_orig.so
for v in rendering_amd/_variants/librtx_*.so; do
  cp $v rendering_amd/librtx_hip.so
  echo "== $(basename $v)"
  RTX_FRAME_MODE=split python tools/shard_stages.py 8 2>&1 | grep -v amdgpu.ids
done
cp /tmp/librtx_orig.so rendering_amd/librtx_hip.so
