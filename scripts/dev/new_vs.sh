#!/bin/bash
# on the GPU box: the in-tree build (NEW) against named A/B builds (scripts/dev/ab_build.sh), in the order given; RAYS = rays per GPU
one() { python bench.py --steps ${STEPS:-20} --warmup 5 --rays ${RAYS:-1024} --no-cpu-baseline --no-extras --scaling weak 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); k=d['kernels']
short={'i2sdf_weight_grads':'wgrad','i2sdf_sdf_backward':'sdf_bwd','i2sdf_sdf_forward_grad':'sdf_fwdg','i2sdf_sample_rays':'sampler','i2sdf_rgb_forward':'rgb_f','i2sdf_rgb_backward':'rgb_b'}
print('$1', 'step', d['ms_per_step'], ' '.join(f'{short[x]}={k[x][\"ms_per_step\"]:.3f}' for x in short if x in k))"; }
for n in "$@"; do
  if [ "$n" = NEW ]; then one NEW; else I2SDF_LIB_PATH=$PWD/i2sdf_amd/lib/ab/libi2sdf_$n.so one $n; fi
done
