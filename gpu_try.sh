#!/bin/bash
# helper: run on the GPU box
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
"$@" > gpurun_out/last.log 2>&1
rc=$?
tail -60 gpurun_out/last.log
exit $rc
