#!/bin/bash
# round 6, experiment 2: attention-phase micro-optimisations (dead / open tiles skip the mask arithmetic, long + short wave per SIMD in the fused text kernel)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r06_exp2; O=gpurun_out/r06_exp2; export TMPDIR=/tmp
t0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_attention.py tests/test_gpu_configs.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest exit $? $(( $(date +%s) - t0 )) s" > $O/status.log
for L in pre_attn attn pre_attn attn; do
  cp plip_amd/csrc/ab/lib_$L.so plip_amd/csrc/libplipmi.so
  echo "=== $L" >> $O/qkvattn.log
  timeout 200 python tools/gpu_diag.py qkvattn 256 77 8 >> $O/qkvattn.log 2>&1
done
cp plip_amd/csrc/ab/lib_attn.so plip_amd/csrc/libplipmi.so
bash tools/ab_libs.sh pre_attn attn > /dev/null 2>&1; cp gpurun_out/ab.log $O/ab.log
echo "all $(( $(date +%s) - t0 )) s" >> $O/status.log
cat $O/status.log; tail -3 $O/pytest.log; grep -E "===|warm|cold|prologue|K loop|images|attention|lifetime|first start" $O/qkvattn.log | cut -c1-200; grep -E "===|one-stream|two-stream" $O/ab.log | cut -c1-260
