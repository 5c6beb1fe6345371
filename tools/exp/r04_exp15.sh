#!/bin/bash
# round-4 experiment 15: L2 touch of the tile after the one being filled (PF 1: activation rows, PF 2: weight rows too) on the 16x16x32 tiles
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/exp15; O=gpurun_out/exp15
( timeout 900 python -m pytest tests/test_gpu_gemm.py -m gpu -q -x > $O/pytest_gemm.log 2>&1; echo "pytest rc $?" >> $O/status.log )
timeout 400 python tools/gpu_diag.py cold 7 17 10 12 13 11 14 9 15 16 > $O/cold.log 2>&1
timeout 900 python tools/gpu_diag.py stepab base "2>10,3>11,6>9" "2>12,3>14,6>15" "2>13,3>14,6>16" "2>12,3>11,6>9" "2>10,3>14,6>9" "2>10,3>11,6>15" "2>17,3>8,6>9" > $O/stepab.log 2>&1
cat $O/status.log; tail -3 $O/pytest_gemm.log; grep -v amdgpu $O/cold.log | tail -8; tail -11 $O/stepab.log
