cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
timeout 1500 python -m pytest tests -q --tb=line -m gpu --durations=25 2>&1 | grep -v "^$" | tail -50
} 2>&1 | tee gpurun_out/r4_call16.log
