#!/bin/bash
# round 6, experiment 4: LayerNorm statistics requested BEFORE the first tile's LDS-DMA fill and folded behind it (GEMM EPI_*_LN kernels + fused text kernel)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r06_exp4; O=gpurun_out/r06_exp4; export TMPDIR=/tmp
t0=$(date +%s)
cp plip_amd/csrc/ab/lib_${KEEP:-lnsplit}.so plip_amd/csrc/libplipmi.so
timeout 1200 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest exit $? $(( $(date +%s) - t0 )) s" > $O/status.log
: > $O/qkvattn.log
for L in ${LIBS:-attn2 lnsplit attn2 lnsplit}; do
  cp plip_amd/csrc/ab/lib_$L.so plip_amd/csrc/libplipmi.so
  echo "=== $L" >> $O/qkvattn.log
  timeout 200 python tools/gpu_diag.py qkvattn 256 77 8 >> $O/qkvattn.log 2>&1
done
bash tools/ab_libs.sh ${LIBS:-attn2 lnsplit} > /dev/null 2>&1; cp gpurun_out/ab.log $O/ab.log
cp plip_amd/csrc/ab/lib_${KEEP:-lnsplit}.so plip_amd/csrc/libplipmi.so
echo "all $(( $(date +%s) - t0 )) s" >> $O/status.log
cat $O/status.log; tail -3 $O/pytest.log; grep -E "===|cold|prologue|lifetime" $O/qkvattn.log | cut -c1-200; grep -E "===|one-stream|two-stream" $O/ab.log | cut -c1-330
