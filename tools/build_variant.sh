#!/bin/bash
# Build pathtracer_amd/lib/libhippt_NAME.so with extra flags for pt_kernels.hip only: tools/build_variant.sh NAME [-Dflag ...]
set -e
cd "$(dirname "$0")/../pathtracer_amd/csrc"
NAME=$1; shift
# the other objects only: libhippt.so itself must always come from the committed sources via plain `make`
make -s build/pt_wavefront.o build/pt_wavefront_bvh.o build/pt_bvh_build.o build/hippt_api.o build/hippt_multi.o build/host_scene.o >/dev/null
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize "$@" -c pt_kernels.hip -o /tmp/pt_kernels_$NAME.o
hipcc -shared -fPIC --offload-arch=gfx950 -o ../lib/libhippt_$NAME.so /tmp/pt_kernels_$NAME.o build/pt_wavefront.o build/pt_wavefront_bvh.o build/pt_bvh_build.o build/hippt_api.o build/hippt_multi.o build/host_scene.o -lz -ldl -lpthread
echo built ../lib/libhippt_$NAME.so
