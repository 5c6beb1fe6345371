#!/bin/bash
# round-4 experiment 17: serpentine MFMA order (one operand changes per MFMA, also at row / group changes) -- experimental variants 7 / 8 / 9 against 2 / 3 / 6
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/exp17; O=gpurun_out/exp17
( timeout 900 python -m pytest tests/test_gpu_gemm.py -m gpu -q -x > $O/pytest_gemm.log 2>&1; echo "pytest rc $?" >> $O/status.log )
timeout 300 python tools/gpu_diag.py libgemm 2 7 3 8 6 9 > $O/libgemm.log 2>&1
for a in "2 12800 2304 768 0" "7 12800 2304 768 0" "6 12800 768 3072 2" "9 12800 768 3072 2"; do
  timeout 120 python tools/gpu_diag.py gemmtrace $a >> $O/gemmtrace.log 2>&1
done
timeout 600 python tools/gpu_diag.py stepab base "2>7,3>8,6>9" "6>9" "2>7,3>8" > $O/stepab.log 2>&1
cat $O/status.log; tail -3 $O/pytest_gemm.log; grep -E "^variant|main loop" $O/gemmtrace.log; grep -E "per tile" $O/libgemm.log | head -9; tail -7 $O/stepab.log
