#!/bin/bash
# Builds librtx_hip.so variants for A/B runs on the GPU box: tools/build_variants.sh "name1:-DFLAGS" "name2:-DFLAGS" ...
# -> rendering_amd/_variants/librtx_<name>.so (benchmarked by tools/bench_variants.sh / tools/bench_ab.sh); the ISA of every variant is kept
# as build/var/<name>/*.s (tools/spill_metric.py build/var/<name>/rtx_api-hip-amdgcn-amd-amdhsa-gfx950.s)
cd "$(dirname "$0")/.."
mkdir -p rendering_amd/_variants build/var
[ -n "$KEEP" ] || rm -f rendering_amd/_variants/librtx_*.so
for spec in "$@"; do
  name=${spec%%:*}; defs=${spec#*:}
  ( mkdir -p build/var/$name
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -save-temps=obj -Rpass-analysis=kernel-resource-usage $defs \
    -o build/var/$name/librtx.so rendering_amd/csrc/rtx_api.hip 2> build/var/$name.log || { echo "$name FAILED"; tail -5 build/var/$name.log; }
    cp build/var/$name/librtx.so rendering_amd/_variants/librtx_$name.so
    rm -f build/var/$name/*.bc build/var/$name/*.hipi build/var/$name/*.o build/var/$name/*host*.s build/var/$name/*.hipfb
    grep -E 'Function Name|VGPRs:|ScratchSize|Occupancy' build/var/$name.log | sed -e 's/.*usage-analysis..//' -e 's/remark: [^ ]* *//' -e 's/ \[-Rpass.*//' | paste - - - - | grep -E 'Pass1KernelILb0ELb1ELb1' | sed "s/^/$name: /" ) &
done
wait
