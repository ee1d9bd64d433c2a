cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests -q --tb=line -m gpu -x 2>&1 | tail -8
} 2>&1 | tee gpurun_out/r4_call12.log
