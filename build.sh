#!/bin/bash
# Builds the gfx950 C-ABI library (hipcc cross-compiles without a GPU).  -ffp-contract=off is part of the
# numerics contract: HIP contracts a*b+c into FMA by default, the reference build does not.
set -e
cd "$(dirname "$0")"
mkdir -p build
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared \
  -save-temps=obj -Rpass-analysis=kernel-resource-usage $RTX_DEFS \
  -o build/librtx_hip.so rendering_amd/csrc/rtx_api.hip 2> build/hipcc.log || { cat build/hipcc.log; exit 1; }
cp build/librtx_hip.so rendering_amd/librtx_hip.so
grep -E 'Function Name|VGPRs:|TotalSGPRs|ScratchSize|Occupancy' build/hipcc.log | sed -e 's/.*usage-analysis..//' -e 's/remark: [^ ]* *//' -e 's/ \[-Rpass.*//' | paste - - - - - | grep -E 'Pass1|Ssaa|Frame' || true
# host side: C++17 Scene/Options/Object API + loaders + BVH builder + flattener, linked against the C ABI
HOST=rendering_amd/host
HIPINC=/opt/rocm/include
g++ -std=c++17 -O2 -ffp-contract=off -fPIC -shared -pthread -D__HIP_PLATFORM_AMD__ -I$HOST/include -I$HIPINC \
  -o rendering_amd/librendering_host.so $HOST/src/util.cpp $HOST/src/lights.cpp $HOST/src/objects.cpp $HOST/src/scene.cpp $HOST/src/capi.cpp \
  -Lrendering_amd -lrtx_hip -Wl,-rpath,'$ORIGIN'
g++ -std=c++17 -O2 -ffp-contract=off -pthread -D__HIP_PLATFORM_AMD__ -I$HOST/include -I$HIPINC -o rendering_amd/render_amd $HOST/src/main.cpp \
  -Lrendering_amd -lrendering_host -lrtx_hip -Wl,-rpath,'$ORIGIN'
echo "built rendering_amd/librtx_hip.so rendering_amd/librendering_host.so rendering_amd/render_amd"
