#!/bin/bash
# on the GPU box: step time per configuration, spec = LIB:PARTS[:WEIGHTS[:HWQ]]  (LIB = NEW or an A/B build name, WEIGHTS e.g. 5,3 or -,
# HWQ = GPU_MAX_HW_QUEUES):  scripts/ab/parts_ab.sh NEW:0 NEW:2 NEW:2:5,3 NEW:4:-:8
one() { python bench.py --steps ${STEPS:-30} --warmup 5 --rays ${RAYS:-1024} --no-cpu-baseline --no-extras --scaling weak 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); k=d['kernels']
short={'i2sdf_weight_grads':'wgrad','i2sdf_sdf_backward':'sdf_bwd','i2sdf_sdf_forward_grad':'sdf_fwdg','i2sdf_sample_rays':'sampler','i2sdf_rgb_forward':'rgb_f','i2sdf_rgb_backward':'rgb_b'}
print('$1', 'step', d['ms_per_step'], ' '.join(f'{short[x]}={k[x][\"ms_per_step\"]:.3f}' for x in short if x in k), flush=True)"; }
for spec in "$@"; do
  IFS=: read -r n parts wts hwq <<< "$spec"
  export I2SDF_PARTS=$parts
  unset I2SDF_PART_WEIGHTS GPU_MAX_HW_QUEUES
  [ -n "$wts" ] && [ "$wts" != "-" ] && export I2SDF_PART_WEIGHTS=$wts
  [ -n "$hwq" ] && export GPU_MAX_HW_QUEUES=$hwq
  if [ "$n" = NEW ]; then unset I2SDF_LIB_PATH; else export I2SDF_LIB_PATH=$PWD/i2sdf_amd/lib/ab/libi2sdf_$n.so; fi
  one $spec
done
