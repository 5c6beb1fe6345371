#!/bin/bash
# round 6, experiment 1: the residual stream's remainder plane in 8 bits (blocked layout) against the 16-bit exact planes -- parity suite, then
# the two builds interleaved on one box (tools/ab_libs.sh: bench one stream / two streams, twice each)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r06_exp1; O=gpurun_out/r06_exp1; export TMPDIR=/tmp
t0=$(date +%s)
timeout 1200 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest exit $? $(( $(date +%s) - t0 )) s" > $O/status.log
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "full_matrix or leading_text or heavy" 2>&1 | grep -E "cos|err|passed|failed" > $O/pytest_prints.log
bash tools/ab_libs.sh base lo8 > /dev/null 2>&1; cp gpurun_out/ab.log $O/ab.log
echo "all $(( $(date +%s) - t0 )) s" >> $O/status.log
cat $O/status.log; tail -3 $O/pytest_gpu.log; cat $O/pytest_prints.log; grep -E "===|one-stream|two-stream" $O/ab.log | cut -c1-330
