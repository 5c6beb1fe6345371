#!/bin/bash
# round-4 experiment 21: the residual GEMMs touch their tile's residual planes (one dword per 128-byte line) before the K loop, so the epilogue finds them
# on this side of HBM -- two builds of the library, one box, twice
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/exp21; O=gpurun_out/exp21; : > $O/ab.log
cp plip_amd/csrc/libplipmi.so /tmp/lib_keep.so
for rep in 1 2; do for L in base touch; do
  cp plip_amd/csrc/ab/lib_$L.so plip_amd/csrc/libplipmi.so
  echo "=== $L rep $rep" >> $O/ab.log
  for a in "6 12800 768 3072 2" "6 12800 768 768 2"; do timeout 120 python tools/gpu_diag.py gemmtrace $a 2>&1 | grep -E "main loop|epilogue|prologue|^variant" | cut -c1-200 >> $O/ab.log; done
  timeout 200 python tools/gpu_diag.py cold 6 2>&1 | grep -E "out|fc2" | cut -c1-200 >> $O/ab.log
  timeout 300 python bench.py --steps 20 --warmup 3 --overlap 0 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('one-stream', d['value'], d['ms_per_step'], d['windows']['ms_per_step'], [(k['name'][-28:], round(k['ms_per_step'],3)) for k in d['kernels'][:3]])" >> $O/ab.log
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('two-stream', d['value'], d['ms_per_step'], d['windows']['ms_per_step'])" >> $O/ab.log
done; done
cp plip_amd/csrc/ab/lib_touch.so plip_amd/csrc/libplipmi.so
timeout 600 python -m pytest tests/test_gpu_gemm.py -m gpu -q -x 2>&1 | tail -2 >> $O/ab.log
cp /tmp/lib_keep.so plip_amd/csrc/libplipmi.so
cat $O/ab.log
