#!/bin/bash
# GPU call 24 of round 6: smoke(), the N > 1 bench line of two ranks on one GPU (gloo) and the whole GPU suite on the final tree
cd $GRAFT_REPO_ROOT
O=gpurun_out
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" > $O/r6_c24_smoke.log 2>&1; tail -2 $O/r6_c24_smoke.log
timeout 600 python bench.py --gpus 2 --backend gloo --share-gpu --steps 5 --warmup 2 --windows 3 --no-cpu-baseline > $O/r6_two_ranks_one_gpu.log 2>&1
python - <<'PY'
import json
ls=[l for l in open("gpurun_out/r6_two_ranks_one_gpu.log") if l.startswith("{")]
if ls:
    d=json.loads(ls[-1]); print({k:d.get(k) for k in ("n_gpus","ms_per_step","ranks","allreduce_us","exposed_allreduce_ms","step_ms_without_allreduce")})
else:
    print(open("gpurun_out/r6_two_ranks_one_gpu.log").read()[-1500:])
PY
timeout 1500 python -m pytest tests -m gpu -q > $O/r6_c24_suite.log 2>&1
tail -3 $O/r6_c24_suite.log
