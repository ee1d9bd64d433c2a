#!/bin/bash
# GPU call 16 of round 6: the whole GPU suite with I2SDF_OPT_SAVES24 on (conf default) -- the gate for adopting it
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > $O/r6_c16_suite.log 2>&1
tail -15 $O/r6_c16_suite.log
