#!/bin/bash
# round-4 experiment 10: the 16x16x32 MFMA form of the GEMM tiles (variants 7 / 8 / 9) against the 32x32x16 form (2 / 3 / 6)
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/exp10; O=gpurun_out/exp10
( timeout 900 python -m pytest tests/test_gpu_gemm.py -m gpu -q -x > $O/pytest_gemm.log 2>&1; echo "pytest rc $?" >> $O/status.log )
timeout 300 python tools/gpu_diag.py tiles 2 7 3 8 6 9 > $O/tiles.log 2>&1
timeout 300 python tools/gpu_diag.py libgemm 2 7 3 8 6 9 > $O/libgemm.log 2>&1
for a in "2 12800 2304 768 0" "7 12800 2304 768 0" "3 12800 3072 768 1" "8 12800 3072 768 1" "6 12800 768 3072 2" "9 12800 768 3072 2" "6 12800 768 768 2" "9 12800 768 768 2"; do
  timeout 120 python tools/gpu_diag.py gemmtrace $a >> $O/gemmtrace.log 2>&1
done
timeout 600 python tools/gpu_diag.py stepab base "2>7" "3>8" "6>9" "2>7,3>8,6>9" > $O/stepab.log 2>&1
cat $O/status.log; tail -4 $O/pytest_gemm.log; grep -v amdgpu $O/tiles.log; grep -E "^variant|main loop" $O/gemmtrace.log; grep -E "bfloat16|per tile" $O/libgemm.log | head -10; tail -8 $O/stepab.log
