#!/bin/bash
# Build the HIP library for gfx950 in-tree: i2sdf_amd/lib/libi2sdf_hip.so
set -e
cd "$(dirname "$0")"
mkdir -p ../lib ../lib/obj
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -Wno-unused-result"
pids=()
for f in plan.cpp pack.hip mlp_fwd.hip mlp_train.hip render.hip mlp_bwd.hip wgrad.hip sampler.hip loss.hip "$@"; do
  o=../lib/obj/$(basename ${f%.*}).o
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ common.h -nt "$o" ] || [ plan.h -nt "$o" ] || [ mlp_common.h -nt "$o" ] || [ ../../include/i2sdf.h -nt "$o" ]; then
    ( hipcc $FLAGS -x hip -c "$f" -o "$o" ) &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libi2sdf_hip.so ../lib/obj/*.o
echo "built $(cd ../lib && pwd)/libi2sdf_hip.so"
