#!/bin/bash
# round 5, experiment 4: fused text kernel with the fill behind the barrier -- timeline, warm/cold against the two kernels, the step
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r05_exp4; O=gpurun_out/r05_exp4; export TMPDIR=/tmp
t0=$(date +%s)
timeout 600 python -m pytest tests/test_gpu_attention.py -m gpu -q -x -k "fused or engine_runs" > $O/pytest_fused.log 2>&1; echo "pytest exit $? $(( $(date +%s) - t0 )) s" > $O/status.log
timeout 300 python tools/gpu_diag.py qkvattn > $O/qkvattn.log 2>&1
timeout 400 python tools/gpu_diag.py stepab f0 base f0 base > $O/stepab.log 2>&1
echo "stepab $(( $(date +%s) - t0 )) s" >> $O/status.log
cat $O/status.log; tail -3 $O/pytest_fused.log; cat $O/qkvattn.log | grep -v amdgpu.ids; grep -E "one stream|max" $O/stepab.log
