#!/bin/bash
# GPU call 2 of round 6: the new tests, then library A/B in one box (r5 baseline, HEAD, four sweep-2 variants), interleaved twice
cd $GRAFT_REPO_ROOT
O=gpurun_out
python -m pytest tests/test_gpu_edge_cases.py tests/test_gpu_sdf_forward.py tests/test_gpu_sampler.py tests/test_gpu_c_example.py -q -x -s > $O/r6_c2_tests.log 2>&1
tail -3 $O/r6_c2_tests.log
python scripts/ab/r6_time.py implicit > $O/r6_c2_implicit.log 2>&1
: > $O/r6_c2_ab.log
for rep in 1 2; do
  for v in r5 head sw2_a2 sw2_a3 sw2_e1 sw2_a2e1; do
    lib=i2sdf_amd/lib/ab/libi2sdf_$v.so; [ $v = head ] && lib=i2sdf_amd/lib/libi2sdf_hip.so
    I2SDF_SAMPLER_BF16X2=0 I2SDF_LIB_PATH=$PWD/$lib python scripts/ab/r6_time.py $v >> $O/r6_c2_ab.log 2>&1
  done
done
I2SDF_LIB_PATH=$PWD/i2sdf_amd/lib/libi2sdf_hip.so python scripts/ab/r6_time.py head_x2 >> $O/r6_c2_ab.log 2>&1
PACED_R6_ONLY=1 timeout 300 scripts/ubench/mfma_paced > $O/r6_c2_paced.log 2>&1
grep -v amdgpu.ids $O/r6_c2_ab.log | tail -80
