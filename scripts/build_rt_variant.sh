#!/bin/bash
# build_rt_variant.sh NAME "-DRT_X=1 ..." : renderer_amd/lib/variant_NAME.so = the library with k_raytrace.hip compiled with extra flags
# (select it at run time with MI355_RENDER_SO; measurement scripts compare kernel variants that way)
set -e
cd "$(dirname "$0")/../renderer_amd/csrc"
NAME=$1; shift
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -Wall -Wno-unused-function -Wno-unused-variable"
/opt/rocm/bin/hipcc $FLAGS "$@" -c k_raytrace.hip -o /tmp/k_raytrace_$NAME.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../lib/variant_$NAME.so capi.o capi_tree.o capi_streams.o capi_diag.o /tmp/k_raytrace_$NAME.o k_raster.o k_points.o k_bvh.o k_post.o k_wire.o mgpu.o -L/opt/rocm/lib -lrccl
echo built variant_$NAME.so
