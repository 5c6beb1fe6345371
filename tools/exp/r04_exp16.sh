#!/bin/bash
# round-4 experiment 16: ring tile, the two waves of a SIMD issue their LDS-DMA fills in different steps (experimental variant 7) against variant 6
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/exp16; O=gpurun_out/exp16
( timeout 900 python -m pytest tests/test_gpu_gemm.py -m gpu -q -x > $O/pytest_gemm.log 2>&1; echo "pytest rc $?" >> $O/status.log )
timeout 300 python tools/gpu_diag.py tiles 6 7 > $O/tiles.log 2>&1
timeout 300 python tools/gpu_diag.py cold 6 7 > $O/cold.log 2>&1
for a in "6 12800 768 3072 2" "7 12800 768 3072 2" "6 12800 768 768 2" "7 12800 768 768 2"; do
  timeout 120 python tools/gpu_diag.py gemmtrace $a >> $O/gemmtrace.log 2>&1
done
timeout 600 python tools/gpu_diag.py stepab base "6>7" > $O/stepab.log 2>&1
cat $O/status.log; tail -3 $O/pytest_gemm.log; grep -v amdgpu $O/tiles.log; grep -v amdgpu $O/cold.log | tail -6; grep -E "^variant|main loop" $O/gemmtrace.log; tail -5 $O/stepab.log
