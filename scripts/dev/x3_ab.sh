#!/bin/bash
# on the GPU box: the sdf-only forward (one sampler pass) for A/B builds of mlp_fwd.hip (scripts/dev/ab_build.sh names; NEW = in-tree);
# CHECK=1 runs the bf16x3 forward / sampler parity tests on each build first
for n in "$@"; do
  echo "== $n"
  if [ "$n" = NEW ]; then unset I2SDF_LIB_PATH; else export I2SDF_LIB_PATH=$PWD/i2sdf_amd/lib/ab/libi2sdf_$n.so; fi
  [ -n "$CHECK" ] && timeout 120 python -m pytest tests -q -x -m gpu -k "forward_bf16x3 or sampler" 2>&1 | tail -1
  ONLY_X3=1 python scripts/dev/x3_time.py 2>/dev/null | tail -2
done
