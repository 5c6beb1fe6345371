#!/bin/bash
# round 5, experiment 5: two-stage tiles with the fill behind the barrier (tiles 8 / 9 against 2 / 3): parity, warm / cold, timelines, the step
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r05_exp5; O=gpurun_out/r05_exp5; export TMPDIR=/tmp
t0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_gemm.py -m gpu -q -x > $O/pytest_gemm.log 2>&1; echo "pytest exit $? $(( $(date +%s) - t0 )) s" > $O/status.log
timeout 300 python tools/gpu_diag.py tiles 2 8 3 9 > $O/tiles.log 2>&1
timeout 300 python tools/gpu_diag.py cold 2 8 3 9 > $O/cold.log 2>&1
for a in "2 12800 2304 768 0" "8 12800 2304 768 0" "3 12800 3072 768 1" "9 12800 3072 768 1"; do timeout 120 python tools/gpu_diag.py gemmtrace $a >> $O/gemmtrace.log 2>&1; done
timeout 400 python tools/gpu_diag.py stepab base "2>8,3>9" "2>8" "3>9" > $O/stepab.log 2>&1
echo "all $(( $(date +%s) - t0 )) s" >> $O/status.log
cat $O/status.log; tail -3 $O/pytest_gemm.log; grep -E "^v\.|^t\." $O/tiles.log $O/cold.log | cut -c1-220; grep -E "^variant|main loop" $O/gemmtrace.log | cut -c1-200; grep -E "one stream|max" $O/stepab.log
