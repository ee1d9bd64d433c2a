#!/bin/bash
# GPU call 15 of round 6: packed 24-bit records of abars / gus / gas (I2SDF_OPT_SAVES24) -- correctness, gradient distance, timing A/B in one process pair
cd $GRAFT_REPO_ROOT
O=gpurun_out
export I2SDF_SAVES24=1
timeout 1200 python -m pytest tests/test_gpu_train_forward.py tests/test_gpu_backward.py tests/test_gpu_determinism.py tests/test_gpu_edge_cases.py -m gpu -q -x > $O/r6_c15_tests1.log 2>&1
tail -6 $O/r6_c15_tests1.log
: > $O/r6_c15_grads.log
I2SDF_SAVES24=0 I2SDF_WGRAD_BF16X2=0 python scripts/ab/r6_emu_grads.py dump prod_x3 >> $O/r6_c15_grads.log 2>&1
I2SDF_SAVES24=0 I2SDF_WGRAD_BF16X2=1 python scripts/ab/r6_emu_grads.py dump prod_x2 >> $O/r6_c15_grads.log 2>&1
I2SDF_SAVES24=1 I2SDF_WGRAD_BF16X2=1 python scripts/ab/r6_emu_grads.py dump s24_x2 >> $O/r6_c15_grads.log 2>&1
python scripts/ab/r6_emu_grads.py compare prod_x3 prod_x2 s24_x2 >> $O/r6_c15_grads.log 2>&1
grep -v "amdgpu\|Warning\|detach\|print" $O/r6_c15_grads.log
rm -f $O/emu_grads_*.pt
: > $O/r6_c15_ab.log
for rep in 1 2; do
  I2SDF_SAVES24=0 python scripts/ab/r6_time.py fp32saves >> $O/r6_c15_ab.log 2>&1
  I2SDF_SAVES24=1 python scripts/ab/r6_time.py saves24 >> $O/r6_c15_ab.log 2>&1
done
grep -v "amdgpu\|Warning\|detach\|print" $O/r6_c15_ab.log
