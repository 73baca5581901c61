#!/bin/bash
# Build pathtracer_amd/lib/libhippt_NAME.so with extra flags for pt_wavefront_bvh.hip only: tools/build_variant_wfb.sh NAME [-Dflag ...]
set -e
cd "$(dirname "$0")/../pathtracer_amd/csrc"
NAME=$1; shift
make -s build/pt_kernels.o build/pt_wavefront.o build/pt_bvh_build.o build/hippt_api.o build/hippt_multi.o build/host_scene.o >/dev/null
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize "$@" -c pt_wavefront_bvh.hip -o /tmp/pt_wfb_$NAME.o
hipcc -shared -fPIC --offload-arch=gfx950 -o ../lib/libhippt_$NAME.so build/pt_kernels.o build/pt_wavefront.o /tmp/pt_wfb_$NAME.o build/pt_bvh_build.o build/hippt_api.o build/hippt_multi.o build/host_scene.o -lz -ldl -lpthread
echo built ../lib/libhippt_$NAME.so
