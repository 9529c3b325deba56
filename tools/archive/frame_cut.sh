cd ${GRAFT_REPO_ROOT:-.}
cp rendering_amd/librtx_hip.so /tmp/orig.so
for c in 3 2 1; do cp rendering_amd/_variants/librtx_cut$c.so rendering_amd/librtx_hip.so; echo "CUT $c"; timeout 200 python tools/frame_check.py scenes/cfg2_smooth_250k.scene 4096 4096 scenes/cfg4_textured_1024.scene 4096 4096 2>&1 | grep -E "scene "; done
cp /tmp/orig.so rendering_amd/librtx_hip.so
