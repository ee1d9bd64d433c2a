cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
run() { timeout 200 python scripts/ab/r4_time.py step 1024 c=2 2>&1 | grep "round 2"; timeout 100 python scripts/ab/r4_time.py fwd 2>&1 | tail -1; }
for rep in 1 2 3; do
  echo "== head"; I2SDF_LIB_PATH=$GRAFT_REPO_ROOT/i2sdf_amd/lib/ab/libi2sdf_head.so run
  echo "== x3hcarry"; I2SDF_LIB_PATH=$GRAFT_REPO_ROOT/i2sdf_amd/lib/ab/libi2sdf_x3hcarry.so run
  echo "== all (carry + prefetch hooks + drain)"; run
done
} 2>&1 | tee gpurun_out/r4_call33.log
