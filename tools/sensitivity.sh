#!/bin/bash
# How sensitive is the render kernel to extra VALU work / extra node fetches?  Builds variants of librt_amd.so with dummy
# instructions (-DRT_EXP_*) into tools/sens/ (git-ignored), to be timed with bench.py through RT_AMD_LIB (binary 48-byte nodes = the C3
# default).  CPU only (cross-compile).  Round-2 result on C3 (ms per step): base 15.97, +12 VALU per node step 16.57, +24: 17.23,
# +1 fetch per node step 18.63, +80 VALU per shade step 16.22, +160: 16.57.  8-ary kernel on C4: base 100.5, +24 VALU 104.3, +48: 105.6, +1 fetch 119.4.
cd "$(dirname "$0")/../ray-tracing_amd" || exit 1
mkdir -p ../tools/sens
for v in "base:" "nv12:-DRT_EXP_NODE_VALU=12" "nv24:-DRT_EXP_NODE_VALU=24" "nv48:-DRT_EXP_NODE_VALU=48" "nl1:-DRT_EXP_NODE_LOAD=1" "sv80:-DRT_EXP_SHADE_VALU=80" "sv160:-DRT_EXP_SHADE_VALU=160"; do
  name=${v%%:*}; flag=${v#*:}
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function --offload-arch=gfx950 \
    -fhip-fp32-correctly-rounded-divide-sqrt -Xarch_host -mfma $flag -DRT_TU_PHASED -fno-slp-vectorize -mllvm -structurizecfg-skip-uniform-regions=1 \
    -c csrc/rt_kernels.hip -o ../tools/sens/k_$name.o || exit 1
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../tools/sens/librt_$name.so ../tools/sens/k_$name.o build/rt_kernels_simple.o build/rt_refit.o build/rt_capture.o build/rt_api.o \
    build/scene_loader.o build/bvh_builder.o build/capture.o build/app.o build/image_decode.o build/host_api.o -lpthread -lz || exit 1
done
rm -f ../tools/sens/*.o
ls -la ../tools/sens
