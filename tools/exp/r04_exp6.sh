#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/exp6
timeout 900 python -m pytest tests/test_gpu_attention.py tests/test_gpu_configs.py -m gpu -q -x -s > gpurun_out/exp6/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/exp6/status.log
timeout 300 python tools/gpu_diag.py attn > gpurun_out/exp6/attn.log 2>&1; echo "attn rc $?" >> gpurun_out/exp6/status.log
timeout 300 python tools/gpu_diag.py towerswap > gpurun_out/exp6/towerswap.log 2>&1; echo "towerswap rc $?" >> gpurun_out/exp6/status.log
cat gpurun_out/exp6/status.log; tail -4 gpurun_out/exp6/pytest.log; grep timing gpurun_out/exp6/attn.log; cat gpurun_out/exp6/towerswap.log
