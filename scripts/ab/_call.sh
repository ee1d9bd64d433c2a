cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
export I2SDF_LIB_PATH=$PWD/i2sdf_amd/lib/ab/libi2sdf_ch2048.so
timeout 600 python -m pytest tests/test_gpu_backward.py tests/test_gpu_determinism.py -q --tb=line 2>&1 | tail -3
for rep in 1 2; do
for v in NEW ch2048; do
  if [ $v = NEW ]; then unset I2SDF_LIB_PATH; else export I2SDF_LIB_PATH=$PWD/i2sdf_amd/lib/ab/libi2sdf_$v.so; fi
  echo "== $v"; timeout 300 python scripts/ab/r4_time.py step 1024 4096 c=2 2>&1 | grep "round [12]"; timeout 300 python scripts/ab/r4_time.py entries 2>&1 | grep "round 1"
done
done
} 2>&1 | tee gpurun_out/r4_call22.log
