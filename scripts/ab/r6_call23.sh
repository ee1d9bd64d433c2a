#!/bin/bash
# GPU call 23 of round 6: loads of the packed records in sweep 2 non-temporal (whole contiguous lines per instruction now) vs plain (shipped)
cd $GRAFT_REPO_ROOT
O=gpurun_out
V=p24ldnt
I2SDF_LIB_PATH=$PWD/i2sdf_amd/lib/ab/libi2sdf_$V.so timeout 600 python -m pytest tests/test_gpu_backward.py tests/test_gpu_saves24.py -m gpu -q -x > $O/r6_c23_tests.log 2>&1; tail -2 $O/r6_c23_tests.log
: > $O/r6_c23_ab.log
for rep in 1 2 3; do
  python scripts/ab/r6_time.py plain >> $O/r6_c23_ab.log 2>&1
  I2SDF_LIB_PATH=$PWD/i2sdf_amd/lib/ab/libi2sdf_$V.so python scripts/ab/r6_time.py nt >> $O/r6_c23_ab.log 2>&1
done
grep "entries\|step round 2" $O/r6_c23_ab.log
