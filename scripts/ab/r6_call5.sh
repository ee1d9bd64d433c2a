#!/bin/bash
# GPU call 5 of round 6: the whole GPU suite on the round's tree, the model kernel on zero / random data, the default bench command
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 900 python -m pytest tests -m gpu -q > $O/r6_c5_suite.log 2>&1
tail -4 $O/r6_c5_suite.log
PACED_R6_ONLY=1 timeout 200 scripts/ubench/mfma_paced > $O/r6_c5_paced_zero.log 2>&1
PACED_R6_ONLY=1 PACED_RANDOM=1 timeout 200 scripts/ubench/mfma_paced > $O/r6_c5_paced_random.log 2>&1
(cd /tmp; export TMPDIR=/tmp; PACED_R6_ONLY=1 PACED_RANDOM=1 timeout 400 python $GRAFT_REPO_ROOT/scripts/ab/pmc_run.py sq valu_kernel staged_kernel -- $GRAFT_REPO_ROOT/scripts/ubench/mfma_paced) > $O/r6_c5_paced_random_pmc.log 2>&1
grep -h "sweep" $O/r6_c5_paced_zero.log $O/r6_c5_paced_random.log; grep -v amdgpu $O/r6_c5_paced_random_pmc.log | cut -c1-300
timeout 600 python bench.py > $O/r6_c5_bench.json 2> $O/r6_c5_bench.err
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r6_c5_bench.json") if l.startswith("{")][-1])
print({k: d[k] for k in ("value","ms_per_step","dtype")})
print("roofline", {k: d["roofline"][k] for k in ("kernel","achieved","peak","frac","traffic","launch_ms")})
for k in ("dense128","wgrad_bf16x3","k1","k5","natural_k","rays4096","cfg3"): print(k, d[k]["ms_per_step"], d[k]["value"])
print("sampler_bf16x3", {k:(v.get("ms_per_step") or v.get("s_per_image")) for k,v in d["sampler_bf16x3"].items()})
print("cfg4_image", d["cfg4_image"]["s_per_image"], "step_hbm_bytes", d.get("step_hbm_bytes"))
print("entry_points", d["roofline"]["entry_points"])
print("cpu", d["cpu_baseline"]["value"], "eager", d["eager_rocm_baseline"].get("ms_per_step"))
PY
