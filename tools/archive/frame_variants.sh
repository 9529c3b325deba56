#!/bin/bash
export RTX_ALLOW_ENV_KNOBS=1      # the product ignores RTX_* environment knobs without it (rtx_api.hip readKnobs)
# GPU box: every variant under rendering_amd/_variants, the frame in one launch, on several workloads (twice, interleaved)
cd ${GRAFT_REPO_ROOT:-.}
cp rendering_amd/librtx_hip.so /tmp/librtx_orig.so
export RTX_FRAME_MODE=fused
for rep in 1 2; do
for v in rendering_amd/_variants/librtx_*.so; do
  cp $v rendering_amd/librtx_hip.so
  echo "== $(basename $v)"
  timeout 300 python tools/frame_check.py ${CASES:-scenes/cfg1_simple_shapes.scene 512 512 scenes/cfg2_smooth_250k.scene 512 512 scenes/cfg2_smooth_250k.scene 1920 1080 scenes/cfg2_smooth_250k.scene 4096 4096 scenes/cfg3_reflective_refractive.scene 1920 1080} 2>&1 | grep scene | sed -e 's/ (modes.*identical/ identical/' | cut -c1-150
done; done
cp /tmp/librtx_orig.so rendering_amd/librtx_hip.so
