#!/bin/bash
# Builds the committed sources (git HEAD, or $1) as rendering_amd/_variants/librtx_base.so: the baseline of an A/B against the working tree.
cd "$(dirname "$0")/.."
REV=${1:-HEAD}
rm -rf /tmp/wt_base && git worktree add -q /tmp/wt_base $REV || exit 1
mkdir -p rendering_amd/_variants
( cd /tmp/wt_base && hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared $BASE_DEFS -o "$OLDPWD/rendering_amd/_variants/librtx_base.so" rendering_amd/csrc/rtx_api.hip 2>&1 | grep -E " error" )
git worktree remove --force /tmp/wt_base
ls -la rendering_amd/_variants/librtx_base.so
