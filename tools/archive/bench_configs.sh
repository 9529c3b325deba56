#!/bin/bash
# On the GPU box: one bench line per BASELINE config (single GPU), for DESIGN.md / profiles.
cd ${GRAFT_REPO_ROOT:-.}
run() { python bench.py --no-cpu-baseline --steps 5 --warmup 1 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print(c['workload'], '|', d['value'], 'Mrays/s', d['ms_per_step'], 'ms/frame', c['frame'], ' pass1', c['pass1_ms'], 'ssaa', c['ssaa_ms'], 'frame kernel', c['frame_kernel_ms'], 'rays', c['rays_per_frame'])"; }
run --scene scenes/cfg1_simple_shapes.scene --width 512 --height 512
run --scene scenes/cfg1_simple_shapes.scene --width 1920 --height 1080
run --scene scenes/cfg2_smooth_250k.scene --width 1920 --height 1080
run --scene scenes/cfg3_reflective_refractive.scene --width 1920 --height 1080
run --scene scenes/cfg4_textured_1024.scene --width 4096 --height 4096
run --scene scenes/cfg2_smooth_250k.scene --width 4096 --height 4096
run --scene scenes/cfg2_smooth_250k.scene --width 8192 --height 8192
