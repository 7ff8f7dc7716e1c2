#!/bin/bash
# usage: scripts/build_variant.sh NAME "-DFOO=1 ..."  -> gpurun_variants/libdocodo_hip_NAME.so (A/B experiments)
set -e
cd "$(dirname "$0")/../docodo_amd/csrc"
mkdir -p ../../variants/obj_$1
for f in *.hip *.cpp; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $2 -x hip -c $f -o ../../variants/obj_$1/${f%.*}.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../variants/libdocodo_hip_$1.so ../../variants/obj_$1/*.o -ldl -lrt -lpthread
rm -rf ../../variants/obj_$1
