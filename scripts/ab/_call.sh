cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_determinism.py tests/test_gpu_baseline_sizes.py -q --tb=line -x 2>&1 | tail -4
timeout 300 python scripts/ab/r4_time.py step 1024 c=2 2>&1 | grep "round [12]"; timeout 300 python scripts/ab/r4_time.py entries 2>&1 | grep "round"
} 2>&1 | tee gpurun_out/r4_call23.log
