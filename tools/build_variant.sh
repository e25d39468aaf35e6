#!/bin/bash
# tools/build_variant.sh NAME [extra hipcc flags...] -> nrays_amd/lib/ab/NAME.so (tuning builds for A/B runs: tools/kbench.py --libs)
set -e
cd "$(dirname "$0")/../nrays_amd/csrc"
name=$1; shift
mkdir -p ../lib/ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -munsafe-fp-atomics -DNR_ONLY_MESH "$@" \
  -o ../lib/ab/$name.so nrays_hip.hip scene_build.cpp bvh_build.cpp multi_gpu.cpp -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
