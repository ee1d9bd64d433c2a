#!/bin/bash
# GPU call 18 of round 6: the round's profile set, the step timeline and the bench line on the tree with the packed 24-bit records
cd $GRAFT_REPO_ROOT
O=gpurun_out
bash scripts/profile_round.sh r6 > $O/r6_c18_profile.log 2>&1
(cd /tmp; export TMPDIR=/tmp; python $GRAFT_REPO_ROOT/scripts/ab/timeline_gaps.py) > $O/r6_step_timeline.txt 2>&1
head -12 $O/r6_step_timeline.txt
timeout 900 python bench.py > $O/r6_bench_line.json 2> $O/r6_bench.err
tail -c 300 $O/r6_bench_line.json; tail -3 $O/r6_bench.err
