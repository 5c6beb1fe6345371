#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/exp9
timeout 600 python -m pytest tests/test_gpu_api.py tests/test_gpu_dropin.py tests/test_reference_loop.py -m gpu -q -x > gpurun_out/exp9/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/exp9/status.log
timeout 300 python tools/gpu_diag.py zeroshot > gpurun_out/exp9/zeroshot.log 2>&1
cat gpurun_out/exp9/status.log; tail -3 gpurun_out/exp9/pytest.log; tail -3 gpurun_out/exp9/zeroshot.log
