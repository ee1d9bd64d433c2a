#!/bin/bash
# GPU call 21 of round 6: sweep 2 with two / three k-chunks of load-ahead again, now that it moves packed records and is no longer bound by its bytes
cd $GRAFT_REPO_ROOT
O=gpurun_out
: > $O/r6_c21_ab.log
for rep in 1 2; do
  python scripts/ab/r6_time.py ahead1 >> $O/r6_c21_ab.log 2>&1
  for v in sw2a2 sw2a3; do
    I2SDF_LIB_PATH=$PWD/i2sdf_amd/lib/ab/libi2sdf_$v.so python scripts/ab/r6_time.py $v >> $O/r6_c21_ab.log 2>&1
  done
done
grep "entries\|step round 2" $O/r6_c21_ab.log
