#!/bin/bash
# GPU box: the overhead probe (real frame / mesh behind the camera) for every variant
cd ${GRAFT_REPO_ROOT:-.}
cp rendering_amd/librtx_hip.so /tmp/librtx_orig.so
for v in rendering_amd/_variants/librtx_*.so; do
  cp $v rendering_amd/librtx_hip.so; echo "== $(basename $v)"
  python tools/overhead_probe.py 2>&1 | grep -E "pass1"
done
cp /tmp/librtx_orig.so rendering_amd/librtx_hip.so
