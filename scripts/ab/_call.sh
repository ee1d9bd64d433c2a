cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
T0=$(date +%s); timeout 600 python -m pytest tests/test_gpu_network.py tests/test_gpu_training_parity.py -q -s -k "full_width_train or training_curves" 2>&1 | grep -E "grad |conditioning|ReLU|passed|failed|Error|PSNR" | cut -c1-250; echo "wall $(( $(date +%s) - T0 )) s"
} 2>&1 | tee gpurun_out/r4_call30.log
