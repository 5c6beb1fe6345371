#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/exp8; O=gpurun_out/exp8
( timeout 600 python -m pytest tests/test_gpu_gemm.py -m gpu -q -x > $O/pytest_gemm.log 2>&1; echo "pytest rc $?" >> $O/status.log )
timeout 300 python tools/gpu_diag.py tiles 2 7 8 3 > $O/tiles.log 2>&1
timeout 300 python tools/gpu_diag.py libgemm 2 7 8 3 > $O/libgemm.log 2>&1
for a in "2 12800 2304 768 0" "7 12800 2304 768 0" "8 12800 2304 768 0" "2 8192 8192 8192 0" "7 8192 8192 8192 0" "8 8192 8192 8192 0"; do
  timeout 120 python tools/gpu_diag.py gemmtrace $a >> $O/gemmtrace.log 2>&1
done
timeout 600 python tools/gpu_diag.py stepab base "2>7" "2>8" > $O/stepab.log 2>&1
cat $O/status.log; tail -2 $O/pytest_gemm.log; grep -v amdgpu $O/tiles.log; grep -E "^variant|main loop" $O/gemmtrace.log; grep -E "torch.bfloat16|per tile" $O/libgemm.log | head -11; tail -6 $O/stepab.log
