#!/bin/bash
# round-4 experiment 1: hand-placed K steps / one wave per SIMD -- correctness, per-kernel time, K-loop timeline, step A/B
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/exp1; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_gemm.py -m gpu -q -x > $O/pytest_gemm.log 2>&1; echo "pytest rc $?" >> $O/status.log )
timeout 300 python tools/gpu_diag.py tiles 2 7 8 14 12 3 9 13 6 10 11 5 > $O/tiles.log 2>&1; echo "tiles rc $?" >> $O/status.log
timeout 300 python tools/gpu_diag.py libgemm 2 7 8 14 12 3 9 6 10 > $O/libgemm.log 2>&1; echo "libgemm rc $?" >> $O/status.log
for a in "2 12800 2304 768 0" "7 12800 2304 768 0" "8 12800 2304 768 0" "14 12800 2304 768 0" "12 12800 2304 768 0" \
         "3 12800 3072 768 1" "9 12800 3072 768 1" "13 12800 3072 768 1" \
         "6 12800 768 3072 2" "10 12800 768 3072 2" "11 12800 768 3072 2" "6 12800 768 768 2" "10 12800 768 768 2" \
         "2 8192 8192 8192 0" "7 8192 8192 8192 0" "12 8192 8192 8192 0"; do
  timeout 120 python tools/gpu_diag.py gemmtrace $a >> $O/gemmtrace.log 2>&1
done; echo "trace done" >> $O/status.log
timeout 600 python tools/gpu_diag.py stepab base "2>7" "2>8" "2>14" "2>12" "3>9" "3>13" "6>10" "6>11" "2>7,3>9,6>10" "2>12,3>9,6>10" > $O/stepab.log 2>&1; echo "stepab rc $?" >> $O/status.log
cat $O/status.log; tail -3 $O/pytest_gemm.log; cat $O/stepab.log | tail -15
