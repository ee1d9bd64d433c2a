cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
timeout 1200 python -m pytest tests/test_gpu_sampler.py tests/test_gpu_loss.py tests/test_gpu_network.py tests/test_gpu_training_parity.py tests/test_gpu_dp_equivalence.py tests/test_gpu_render_image.py tests/test_gpu_edge_cases.py -q --tb=short -x 2>&1 | grep -v "^$" | tail -15
timeout 300 python scripts/ab/r4_time.py step 1024 c=2 2>&1 | grep -v amdgpu
timeout 600 python scripts/ab/timeline_gaps.py | tail -52
} 2>&1 | tee gpurun_out/r4_call15.log
