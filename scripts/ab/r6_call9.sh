#!/bin/bash
# GPU call 9 of round 6: the sampler's wave scans by DPP instead of ds_bpermute -- parity tests, then A/B against the shuffle build
cd $GRAFT_REPO_ROOT
O=gpurun_out
python -m pytest tests/test_gpu_sampler.py tests/test_gpu_network.py tests/test_gpu_baseline_sizes.py tests/test_gpu_full_image.py tests/test_gpu_render_image.py tests/test_gpu_determinism.py -q > $O/r6_c9_tests.log 2>&1
tail -4 $O/r6_c9_tests.log
: > $O/r6_c9_ab.log
for rep in 1 2; do
  I2SDF_LIB_PATH=$PWD/i2sdf_amd/lib/libi2sdf_hip.so python scripts/ab/r6_time.py sampler dpp >> $O/r6_c9_ab.log 2>&1
  I2SDF_LIB_PATH=$PWD/i2sdf_amd/lib/ab/libi2sdf_nodpp.so python scripts/ab/r6_time.py sampler shfl >> $O/r6_c9_ab.log 2>&1
done
grep -v amdgpu $O/r6_c9_ab.log
