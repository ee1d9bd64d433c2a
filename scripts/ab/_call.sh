cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_network.py -q -s --tb=line 2>&1 | grep -E "MEASURED|passed|failed"
timeout 300 python scripts/ab/r4_time.py step 1024 c=2 2>&1 | grep -v amdgpu
} 2>&1 | tee gpurun_out/r4_call18.log
