#!/bin/bash
# Build the HIP library for gfx950 in-tree: i2sdf_amd/lib/libi2sdf_hip.so
set -e
cd "$(dirname "$0")"
mkdir -p ../lib ../lib/obj
# the K-outer bf16x3 ops (x3.h) are fully unrolled stage loops of several thousand instructions: lift the pragma-unroll cap,
# otherwise the loop stays rolled, its register arrays are indexed dynamically and land in scratch
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -Wno-unused-result"
X3FLAGS="-mllvm -pragma-unroll-threshold=1000000"     # only where the bf16x3 training kernels live
pids=()
for f in plan.cpp comm.cpp render_image.cpp pack.hip mlp_fwd.hip mlp_train.hip mlp_x3.hip mlp_x3p.hip mlp_x3h.hip render.hip mlp_bwd.hip wgrad.hip sampler.hip loss.hip optim.hip grid.hip draws.hip "$@"; do
  o=../lib/obj/$(basename ${f%.*}).o
  stale=0
  for h in "$f" *.h ../../include/i2sdf.h build.sh; do [ "$h" -nt "$o" ] && stale=1; done
  [ "$f" = mlp_x3p.hip ] && [ mlp_x3.hip -nt "$o" ] && stale=1          # (mlp_x3p.hip is mlp_x3.hip with I2SDF_X3_P24_TU)
  if [ ! -f "$o" ] || [ $stale = 1 ]; then
    extra=""; [ "$f" = mlp_x3.hip ] && extra="$X3FLAGS"
    [ "$f" = mlp_x3p.hip ] && extra="$X3FLAGS"          # (the packing instantiations of mlp_x3.hip's kernels: same flags)
    # wgrad.hip: the hand-placed stage of wgrad3p_body is a 96-unit unrolled loop whose scalar ops must stay where they are written
    [ "$f" = wgrad.hip ] && extra="$X3FLAGS -fno-slp-vectorize"
    # mlp_x3h.hip: one fenced unit per MFMA (x3h.h); packed f32 VALU ops beside MFMAs are slower than the scalar ones they replace
    [ "$f" = mlp_x3h.hip ] && extra="$X3FLAGS -fno-slp-vectorize"
    ( hipcc $FLAGS $extra -x hip -c "$f" -o "$o" ) &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libi2sdf_hip.so ../lib/obj/*.o
echo "built $(cd ../lib && pwd)/libi2sdf_hip.so"
