#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/exp7
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/exp7/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/exp7/status.log
timeout 300 python tools/gpu_diag.py latency > gpurun_out/exp7/latency.log 2>&1; echo "latency rc $?" >> gpurun_out/exp7/status.log
timeout 300 python tools/gpu_diag.py attn > gpurun_out/exp7/attn.log 2>&1
timeout 300 python tools/gpu_diag.py towerswap > gpurun_out/exp7/towerswap.log 2>&1
cat gpurun_out/exp7/status.log; grep -E "^FAILED|passed|failed" gpurun_out/exp7/pytest.log | tail -40; cat gpurun_out/exp7/latency.log; grep "timing.*impl 1" gpurun_out/exp7/attn.log; grep -E "^ViT|attention" gpurun_out/exp7/towerswap.log
