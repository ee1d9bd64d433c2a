cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
timeout 300 python scripts/ab/glue_profile.py 2>&1 | tail -60
timeout 300 python -m pytest tests/test_gpu_loss.py tests/test_gpu_training_parity.py -q -x 2>&1 | tail -3
} 2>&1 | tee gpurun_out/r4_call36.log
