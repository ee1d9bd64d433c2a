#!/bin/bash
# A/B build of SEVERAL source files with extra compiler flags:
#   scripts/ab/variant_build.sh NAME "EXTRA FLAGS" file1.hip file2.hip ...
# -> i2sdf_amd/lib/ab/libi2sdf_NAME.so (the other objects come from the in-tree build; select with I2SDF_LIB_PATH)
set -e
cd "$(dirname "$0")/../../i2sdf_amd/csrc"
name=$1; extra=$2; shift 2
rm -rf ../lib/ab/$name ../lib/ab/libi2sdf_$name.so; mkdir -p ../lib/ab/$name
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -Wno-unused-result"
X3FLAGS="-mllvm -pragma-unroll-threshold=1000000"
skip=""
pids=()
for f in "$@"; do
  base=$(basename ${f%.*})
  skip="$skip /$base.o"
  fx=""; [ "$f" = mlp_x3.hip ] && fx="$X3FLAGS"
  [ "$f" = mlp_x3p.hip ] && fx="$X3FLAGS"
  [ "$f" = mlp_x3h.hip ] && fx="$X3FLAGS -fno-slp-vectorize"
  [ "$f" = wgrad.hip ] && fx="$X3FLAGS -fno-slp-vectorize"
  ( hipcc $FLAGS $fx $extra -x hip -c "$f" -o ../lib/ab/$name/$base.o ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
others=""
for o in ../lib/obj/*.o; do
  keep=1; for s in $skip; do [[ "$o" == *"$s" ]] && keep=0; done
  [ $keep = 1 ] && others="$others $o"
done
hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/ab/libi2sdf_$name.so ../lib/ab/$name/*.o $others
echo "built $name"
