#!/bin/bash
# round 5, experiment 7: the ring tile with the STREAMED K loop of the two-stage tiles (experimental tile 8) against tile 6 (fully double-buffered fragments)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r05_exp7; O=gpurun_out/r05_exp7; export TMPDIR=/tmp
t0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_gemm.py -m gpu -q -x > $O/pytest_gemm.log 2>&1; echo "pytest exit $? $(( $(date +%s) - t0 )) s" > $O/status.log
timeout 300 python tools/gpu_diag.py tiles 6 8 > $O/tiles.log 2>&1
timeout 300 python tools/gpu_diag.py cold 6 8 > $O/cold.log 2>&1
for a in "6 12800 768 3072 2" "8 12800 768 3072 2" "6 12800 768 768 2" "8 12800 768 768 2" "6 19712 512 2048 2" "8 19712 512 2048 2"; do timeout 120 python tools/gpu_diag.py gemmtrace $a >> $O/gemmtrace.log 2>&1; done
timeout 400 python tools/gpu_diag.py stepab base "6>8" base "6>8" > $O/stepab.log 2>&1
echo "all $(( $(date +%s) - t0 )) s" >> $O/status.log
cat $O/status.log; tail -3 $O/pytest_gemm.log; grep -E "^v\.|^t\." $O/tiles.log $O/cold.log | cut -c1-200; grep -E "^variant|main loop|epilogue" $O/gemmtrace.log | cut -c1-200; grep -E "one stream|max" $O/stepab.log
