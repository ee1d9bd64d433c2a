#!/bin/bash
# GPU call 6 of round 6: the whole GPU suite on the final tree, the round's profile set, the step timeline, the bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > $O/r6_c6_suite.log 2>&1
tail -4 $O/r6_c6_suite.log
bash scripts/profile_round.sh r6 > $O/r6_c6_profile.log 2>&1
(cd /tmp; export TMPDIR=/tmp; python $GRAFT_REPO_ROOT/scripts/ab/timeline_gaps.py) > $O/r6_step_timeline.txt 2>&1
head -12 $O/r6_step_timeline.txt
timeout 900 python bench.py > $O/r6_bench_line.json 2> $O/r6_bench.err
tail -c 600 $O/r6_bench_line.json
