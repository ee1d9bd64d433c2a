#!/bin/bash
# GPU call 20 of round 6: sweep 2 with e = exp(-100 h) shared between sigma and 1 - sigma (X3_SW2_SHARE_E=1), now that the sweep is no longer bound by its bytes
cd $GRAFT_REPO_ROOT
O=gpurun_out
V=${1:-sw2e}
I2SDF_LIB_PATH=$PWD/i2sdf_amd/lib/ab/libi2sdf_$V.so timeout 600 python -m pytest tests/test_gpu_backward.py tests/test_gpu_saves24.py -m gpu -q -x > $O/r6_c20_tests.log 2>&1; tail -2 $O/r6_c20_tests.log
: > $O/r6_c20_ab.log
for rep in 1 2 3; do
  python scripts/ab/r6_time.py head >> $O/r6_c20_ab.log 2>&1
  I2SDF_LIB_PATH=$PWD/i2sdf_amd/lib/ab/libi2sdf_$V.so python scripts/ab/r6_time.py $V >> $O/r6_c20_ab.log 2>&1
done
grep "entries\|step round 2" $O/r6_c20_ab.log
