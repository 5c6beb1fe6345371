#!/bin/bash
# round-4 experiment 19: 320x256 tile with FOUR instead of two MFMA groups behind its barrier (two builds of the library, one box)
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/exp19; O=gpurun_out/exp19; : > $O/ab.log
cp plip_amd/csrc/libplipmi.so /tmp/lib_keep.so
for rep in 1 2; do for L in pb2 pb4; do
  cp plip_amd/csrc/ab/lib_$L.so plip_amd/csrc/libplipmi.so
  echo "=== $L rep $rep" >> $O/ab.log
  timeout 120 python tools/gpu_diag.py gemmtrace 3 12800 3072 768 1 2>&1 | grep -E "main loop|^variant" | cut -c1-200 >> $O/ab.log
  timeout 200 python tools/gpu_diag.py tiles 3 2>&1 | grep -E "fc1" >> $O/ab.log
  timeout 200 python tools/gpu_diag.py libgemm 3 2>&1 | grep -E "fc1|big" | grep bfloat16 -A0 | head -4 >> $O/ab.log
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('two-stream', d['value'], d['ms_per_step'], d['windows']['ms_per_step'])" >> $O/ab.log
done; done
cp /tmp/lib_keep.so plip_amd/csrc/libplipmi.so
timeout 600 python -m pytest tests/test_gpu_gemm.py -m gpu -q -x 2>&1 | tail -2 >> $O/ab.log
cat $O/ab.log
