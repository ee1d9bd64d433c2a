#!/bin/bash
# GPU call 3 of round 6: where do the real sweeps differ from the model kernel that has the same bytes, MFMAs and VALU mix?  PMC sets on both.
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6_c3_pmc.txt
: > $O
BARGS="--no-cpu-baseline --no-extras --scaling weak --windows 1 --profile-steps 0 --no-live-traffic"
for set in sq icache lds vmem l2 mem memw; do
  echo "== set $set: model kernels (scripts/ubench/mfma_paced, round-6 section)" >> $O
  PACED_R6_ONLY=1 python $R/scripts/ab/pmc_run.py $set valu_kernel staged_kernel -- $R/scripts/ubench/mfma_paced >> $O 2>&1
  echo "== set $set: the real kernels (bench.py headline steps, I2SDF_SAMPLER_BF16X2 default)" >> $O
  (cd $R && python scripts/ab/pmc_run.py $set sweep igrad wgrad3p sdf_fwd3h sdf_train -- python bench.py --steps 2 --warmup 1 $BARGS) >> $O 2>&1
done
cat $O
