#!/bin/bash
# GPU call 13 of round 6: numerical gate by emulation for narrower storage of G(a) / abar / G(hbar) (VERDICT r5 task 2b)
cd $GRAFT_REPO_ROOT
O=gpurun_out
L=i2sdf_amd/lib/ab
: > $O/r6_c13_emu.log
I2SDF_WGRAD_BF16X2=0 python scripts/ab/r6_emu_grads.py dump prod_x3 >> $O/r6_c13_emu.log 2>&1
I2SDF_WGRAD_BF16X2=1 python scripts/ab/r6_emu_grads.py dump prod_x2 >> $O/r6_c13_emu.log 2>&1
for v in ga8 ga13 ga16 all8 all13; do
  I2SDF_LIB_PATH=$PWD/$L/libi2sdf_$v.so I2SDF_WGRAD_BF16X2=1 python scripts/ab/r6_emu_grads.py dump ${v}_x2 >> $O/r6_c13_emu.log 2>&1
done
I2SDF_LIB_PATH=$PWD/$L/libi2sdf_ga13.so I2SDF_WGRAD_BF16X2=0 python scripts/ab/r6_emu_grads.py dump ga13_x3 >> $O/r6_c13_emu.log 2>&1
python scripts/ab/r6_emu_grads.py compare prod_x3 prod_x2 ga8_x2 ga13_x2 ga16_x2 all8_x2 all13_x2 ga13_x3 >> $O/r6_c13_emu.log 2>&1
grep -v amdgpu $O/r6_c13_emu.log
rm -f $O/emu_grads_*.pt
# the GPU suite (without the ensemble) under the two most interesting builds (both weight-gradient modes via conftest)
for v in ga13 all8; do
  I2SDF_LIB_PATH=$PWD/$L/libi2sdf_$v.so timeout 900 python -m pytest tests -m gpu -q --ignore=tests/test_gpu_psnr_ensemble.py --ignore=tests/test_gpu_c_example.py > $O/r6_c13_tests_$v.log 2>&1
  echo "== $v"; tail -5 $O/r6_c13_tests_$v.log
done
