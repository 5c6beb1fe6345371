#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/exp4
timeout 300 python tools/gpu_diag.py cumask lo128 lo144 lo160 xcd4 xcd5 even > gpurun_out/exp4/cumask.log 2>&1; echo rc $? >> gpurun_out/exp4/cumask.log
cat gpurun_out/exp4/cumask.log
