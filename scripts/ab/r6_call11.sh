#!/bin/bash
# GPU call 11 of round 6: task order inside the merged weight-gradient grid; the training-parity module after the criterion fix
cd $GRAFT_REPO_ROOT
O=gpurun_out
python -m pytest tests/test_gpu_training_parity.py -q > $O/r6_c11_tests.log 2>&1; tail -3 $O/r6_c11_tests.log
: > $O/r6_c11_ab.log
for rep in 1 2 3; do
  python scripts/ab/r6_time.py narrow_first >> $O/r6_c11_ab.log 2>&1
  I2SDF_WGRAD_BLOCKS_FIRST=1 python scripts/ab/r6_time.py blocks_first >> $O/r6_c11_ab.log 2>&1
done
grep -v amdgpu $O/r6_c11_ab.log
