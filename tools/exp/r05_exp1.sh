#!/bin/bash
# round 5, experiment 1: the 160x128 two-per-CU tile (variant 7) -- parity of every epilogue, in-kernel timelines with and
# without per-slot issue priority, product epilogues warm / cold against the round-4 tiles, and the step under remapped choices.
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r05_exp1; O=gpurun_out/r05_exp1; export TMPDIR=/tmp
t0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_gemm.py -m gpu -q -x > $O/pytest_gemm.log 2>&1; echo "pytest_gemm exit $? $(( $(date +%s) - t0 )) s" > $O/status.log
for duo in 0 1 2; do
  for a in "7 12800 768 768 2" "7 12800 768 3072 2" "7 19712 512 512 2" "7 19712 512 2048 2" "7 12800 2304 768 0" "7 12800 3072 768 1"; do
    timeout 120 python tools/gpu_diag.py gemmtrace $a $duo >> $O/gemmtrace_duo$duo.log 2>&1
  done
done
for a in "6 12800 768 768 2" "6 12800 768 3072 2" "2 12800 2304 768 0" "3 12800 3072 768 1"; do
  timeout 120 python tools/gpu_diag.py gemmtrace $a >> $O/gemmtrace_ref.log 2>&1
done
echo "traces $(( $(date +%s) - t0 )) s" >> $O/status.log
timeout 300 python tools/gpu_diag.py tiles 2 3 6 7 > $O/tiles.log 2>&1
timeout 300 python tools/gpu_diag.py cold 2 3 6 7 > $O/cold.log 2>&1
echo "tiles+cold $(( $(date +%s) - t0 )) s" >> $O/status.log
timeout 400 python tools/gpu_diag.py stepab base "6>7" "6>7,d1" "6>7,2>7" "6>7,2>7,3>7" "6>7,2>7,3>7,d1" > $O/stepab.log 2>&1
echo "stepab $(( $(date +%s) - t0 )) s" >> $O/status.log
cat $O/status.log; tail -5 $O/pytest_gemm.log; cat $O/stepab.log | tail -12
