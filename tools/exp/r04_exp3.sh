#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/exp3
timeout 200 tools/exp/mfma_power > gpurun_out/exp3/mfma_power.log 2>&1; echo rc $? >> gpurun_out/exp3/mfma_power.log
cat gpurun_out/exp3/mfma_power.log
