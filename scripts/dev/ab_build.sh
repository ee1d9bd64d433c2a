#!/bin/bash
# A/B builds of ONE source file with different -D flags: scripts/dev/ab_build.sh wgrad.hip name1 "-DX=1" name2 "-DX=2" ...
# -> i2sdf_amd/lib/ab/libi2sdf_<name>.so (travels with gpurun; select with I2SDF_LIB_PATH)
set -e
cd "$(dirname "$0")/../../i2sdf_amd/csrc"
src=$1; shift
mkdir -p ../lib/ab
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -Wno-unused-result -mllvm -pragma-unroll-threshold=1000000"
base=$(basename ${src%.*})
others=$(ls ../lib/obj/*.o | grep -v "/$base.o")
pids=()
while [ $# -gt 0 ]; do
  name=$1; defs=$2; shift 2
  ( hipcc $FLAGS $defs -x hip -c "$src" -o ../lib/ab/$base.$name.o && hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/ab/libi2sdf_$name.so ../lib/ab/$base.$name.o $others && echo "built $name" ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
