#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/exp5
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_host.py tests/test_gpu_api.py -m gpu -q -x -s -k "leading_text or recode or text_tower_f16 or coalesce or out_of_range or host" > gpurun_out/exp5/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/exp5/status.log
timeout 600 python tools/gpu_diag.py mixed 0 2 4 6 8 12 > gpurun_out/exp5/mixed.log 2>&1; echo "mixed rc $?" >> gpurun_out/exp5/status.log
cat gpurun_out/exp5/status.log; grep -E "passed|failed|cos err|Error" gpurun_out/exp5/pytest.log | tail -12; cat gpurun_out/exp5/mixed.log
