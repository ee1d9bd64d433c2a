cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
run() { timeout 200 python scripts/ab/r4_time.py step 1024 c=$1 2>&1 | grep "round 2"; }
for rep in 1 2; do
  echo "== in-tree"; run 2
  echo "== narrow_last"; I2SDF_LIB_PATH=$GRAFT_REPO_ROOT/i2sdf_amd/lib/ab/libi2sdf_narrow_last.so run 2
done
for w in "32,18" "18,32" "34,16" "28,22"; do echo "== weights $w (2 ranges)"; I2SDF_PART_WEIGHTS=$w run 2; done
for w in "16,16,18" "18,16,16" "20,16,14"; do echo "== weights $w (3 ranges)"; I2SDF_PART_WEIGHTS=$w run 3; done
for w in "16,16,16,2" "16,16,14,4"; do echo "== weights $w (4 ranges)"; I2SDF_PART_WEIGHTS=$w run 4; done
} 2>&1 | tee gpurun_out/r4_call28.log
