#!/bin/bash
# build_rs_variant.sh NAME "-DSMT_X=1 ..." : renderer_amd/lib/variant_NAME.so = the library with k_raster.hip compiled with extra flags
# (select it at run time with MI355_RENDER_SO; see build_rt_variant.sh)
set -e
cd "$(dirname "$0")/../renderer_amd/csrc"
NAME=$1; shift
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -Wall -Wno-unused-function -Wno-unused-variable"
/opt/rocm/bin/hipcc $FLAGS "$@" -c k_raster.hip -o /tmp/k_raster_$NAME.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../lib/variant_$NAME.so capi.o capi_tree.o capi_streams.o capi_diag.o k_raytrace.o /tmp/k_raster_$NAME.o k_points.o k_bvh.o k_post.o k_wire.o mgpu.o -L/opt/rocm/lib -lrccl
echo built variant_$NAME.so
