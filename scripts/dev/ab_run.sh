#!/bin/bash
# on the GPU box: step time and per-entry-point times for every variant built by ab_build.sh (names as arguments; repeat a name to
# see the run-to-run spread -- the chip's DVFS moves a 20-step measurement by 1-2 %)
for n in "$@"; do
  I2SDF_LIB_PATH=$PWD/i2sdf_amd/lib/ab/libi2sdf_$n.so python bench.py --steps ${STEPS:-40} --warmup 5 --no-cpu-baseline --no-extras --scaling weak 2>/dev/null | tail -1 | \
    python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); k=d['kernels']
short={'i2sdf_weight_grads':'wgrad','i2sdf_sdf_backward':'sdf_bwd','i2sdf_sdf_forward_grad':'sdf_fwdg','i2sdf_sample_rays':'sampler','i2sdf_rgb_forward':'rgb_f','i2sdf_rgb_backward':'rgb_b'}
print('$n', 'step', d['ms_per_step'], ' '.join(f'{short[x]}={k[x][\"ms_per_step\"]:.3f}' for x in short if x in k))"
done
