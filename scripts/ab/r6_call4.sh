#!/bin/bash
# GPU call 4 of round 6: fused loss + render backward, merged weight-gradient launch, drain look-ahead of the sweeps
cd $GRAFT_REPO_ROOT
O=gpurun_out
python -m pytest tests/test_gpu_loss.py tests/test_gpu_backward.py tests/test_gpu_edge_cases.py -q -x > $O/r6_c4_tests.log 2>&1
tail -15 $O/r6_c4_tests.log
: > $O/r6_c4_ab.log
HEAD=$PWD/i2sdf_amd/lib/libi2sdf_hip.so
for rep in 1 2; do
  I2SDF_LIB_PATH=$HEAD python scripts/ab/r6_time.py head >> $O/r6_c4_ab.log 2>&1
  I2SDF_FUSED_RENDER_LOSS=0 I2SDF_LIB_PATH=$HEAD python scripts/ab/r6_time.py nofuse >> $O/r6_c4_ab.log 2>&1
  I2SDF_WGRAD_MERGED=0 I2SDF_LIB_PATH=$HEAD python scripts/ab/r6_time.py nomerge >> $O/r6_c4_ab.log 2>&1
  for v in sw_d3 sw_d4 sw_d7; do
    I2SDF_LIB_PATH=$PWD/i2sdf_amd/lib/ab/libi2sdf_$v.so python scripts/ab/r6_time.py $v >> $O/r6_c4_ab.log 2>&1
  done
done
grep -v amdgpu.ids $O/r6_c4_ab.log | tail -70
