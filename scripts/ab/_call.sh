cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_baseline_sizes.py tests/test_gpu_batcher.py tests/test_gpu_c_example.py tests/test_gpu_determinism.py -q --tb=short -x 2>&1 | grep -v "^$" | tail -30
} 2>&1 | tee gpurun_out/r4_call14.log
