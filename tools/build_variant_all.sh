#!/bin/bash
# Build pathtracer_amd/lib/libhippt_NAME.so with extra flags for every .hip file: tools/build_variant_all.sh NAME [-Dflag ...]
set -e
cd "$(dirname "$0")/../pathtracer_amd/csrc"
NAME=$1; shift
D=/tmp/hippt_var_$NAME; mkdir -p $D
make -s build/host_scene.o >/dev/null
for f in pt_kernels pt_wavefront pt_wavefront_bvh pt_bvh_build hippt_api hippt_multi; do
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize "$@" -c $f.hip -o $D/$f.o 2>/dev/null &
done
wait
hipcc -shared -fPIC --offload-arch=gfx950 -o ../lib/libhippt_$NAME.so $D/*.o build/host_scene.o -lz -ldl -lpthread
echo built ../lib/libhippt_$NAME.so
