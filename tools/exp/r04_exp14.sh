#!/bin/bash
# round-4 experiment 14: in the step (operands from HBM / the far L2), does the ring's deeper lookahead beat the big two-stage tiles on q/k/v and fc1?
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/exp14; O=gpurun_out/exp14
timeout 900 python tools/gpu_diag.py stepab base "2>6" "3>6" "2>7,3>8,6>9" "2>9,3>8,6>9" "2>7,3>9,6>9" "2>9,3>9,6>9" "2>10,3>11,6>9" > $O/stepab.log 2>&1
timeout 400 python tools/gpu_diag.py cold 2 7 10 3 8 11 6 9 > $O/cold.log 2>&1
tail -11 $O/stepab.log; grep -v amdgpu $O/cold.log | tail -30
