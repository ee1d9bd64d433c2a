#!/bin/bash
# quick A/B on the GPU box: step time + the roofline kernel's average duration (bench.py without the extra legs)
python bench.py --steps ${1:-20} --warmup 5 --no-cpu-baseline --no-extras --scaling weak 2>/dev/null | tail -1 | \
  python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('ms_per_step', d['ms_per_step']); print('roofline', json.dumps(d['roofline']))"
