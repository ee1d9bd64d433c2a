#!/bin/bash
# GPU call 19 of round 6: raw rows of the 256x256 weight-gradient body two stages in flight (W3_RAW_SETS=2) vs one (shipped); correctness of the variant first
cd $GRAFT_REPO_ROOT
O=gpurun_out
I2SDF_LIB_PATH=$PWD/i2sdf_amd/lib/ab/libi2sdf_raw2.so timeout 600 python -m pytest tests/test_gpu_backward.py tests/test_gpu_saves24.py -m gpu -q -x > $O/r6_c19_tests.log 2>&1; tail -2 $O/r6_c19_tests.log
: > $O/r6_c19_ab.log
for rep in 1 2 3; do
  python scripts/ab/r6_time.py rs1 >> $O/r6_c19_ab.log 2>&1
  I2SDF_LIB_PATH=$PWD/i2sdf_amd/lib/ab/libi2sdf_raw2.so python scripts/ab/r6_time.py rs2 >> $O/r6_c19_ab.log 2>&1
done
grep "entries\|step round 2" $O/r6_c19_ab.log
