#!/bin/bash
# round-4 experiment 12: which of the three 16x16x32 tiles pay in the step (repeated arms, one box)
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/exp12; O=gpurun_out/exp12
timeout 900 python tools/gpu_diag.py stepab base "2>7,3>8" "2>7,3>8,6>9" "6>9" base "2>7,3>8" "2>7,3>8,6>9" "6>9" > $O/stepab.log 2>&1
for a in "2 12800 2304 768 0" "7 12800 2304 768 0"; do timeout 120 python tools/gpu_diag.py gemmtrace $a >> $O/gemmtrace.log 2>&1; done
grep -E "^variant|main loop" $O/gemmtrace.log; tail -11 $O/stepab.log
