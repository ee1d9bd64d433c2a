#!/bin/bash
# GPU call 14 of round 6: timing knock-out "24-bit records of abar / G(hbar) / G(a) in the 32-point kernels" against production, interleaved
cd $GRAFT_REPO_ROOT
O=gpurun_out
: > $O/r6_c14_ab.log
for rep in 1 2; do
  python scripts/ab/r6_time.py prod >> $O/r6_c14_ab.log 2>&1
  I2SDF_LIB_PATH=$PWD/i2sdf_amd/lib/ab/libi2sdf_shape24.so python scripts/ab/r6_time.py shape24 >> $O/r6_c14_ab.log 2>&1
done
grep -v "amdgpu\|Warning\|detach\|print" $O/r6_c14_ab.log
