#!/bin/bash
# round 6, experiment 3: fused text kernel, prologue order (tile 1's fills behind the compiler-visible loads)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r06_exp3; O=gpurun_out/r06_exp3; export TMPDIR=/tmp
: > $O/qkvattn.log
for L in ${LIBS:-attn attn2 attn attn2}; do
  cp plip_amd/csrc/ab/lib_$L.so plip_amd/csrc/libplipmi.so
  echo "=== $L" >> $O/qkvattn.log
  timeout 200 python tools/gpu_diag.py qkvattn 256 77 8 >> $O/qkvattn.log 2>&1
done
cp plip_amd/csrc/ab/lib_${KEEP:-attn2}.so plip_amd/csrc/libplipmi.so
timeout 600 python -m pytest tests/test_gpu_attention.py -m gpu -q -x > $O/pytest.log 2>&1; tail -2 $O/pytest.log
grep -E "===|warm|cold|prologue|K loop|images|attention|lifetime" $O/qkvattn.log | cut -c1-200
