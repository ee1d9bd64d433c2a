#!/bin/bash
# GPU call 12 of round 6: does the Infinity Cache carry a producer's output to its consumer?  (mall_probe) and per-point cost of the step by batch width
cd $GRAFT_REPO_ROOT
O=gpurun_out
hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/ubench/mall_probe.hip -o /tmp/mall_probe && timeout 300 /tmp/mall_probe > $O/r6_c12_mall.log 2>&1
cat $O/r6_c12_mall.log
timeout 600 python scripts/ab/r6_small.py > $O/r6_c12_small.log 2>&1
grep -v amdgpu $O/r6_c12_small.log
