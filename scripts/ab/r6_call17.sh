#!/bin/bash
# GPU call 17 of round 6: loads of the packed 24-bit operands in the 256x256 weight-gradient body: plain (shipped) vs non-temporal
cd $GRAFT_REPO_ROOT
O=gpurun_out
: > $O/r6_c17_ab.log
for rep in 1 2 3; do
  python scripts/ab/r6_time.py plain >> $O/r6_c17_ab.log 2>&1
  I2SDF_LIB_PATH=$PWD/i2sdf_amd/lib/ab/libi2sdf_p24nt.so python scripts/ab/r6_time.py nt >> $O/r6_c17_ab.log 2>&1
done
grep -v "amdgpu\|Warning\|detach\|print" $O/r6_c17_ab.log
