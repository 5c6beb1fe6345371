#!/bin/bash
# round-4 experiment 20: two-stage 16x16x32 tiles, the fill of tile kt+2 issued right behind tile kt's barrier (a whole tile of lead) instead of in
# the first groups of tile kt+1 -- two builds of the library, one box, twice
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/exp20; O=gpurun_out/exp20; : > $O/ab.log
cp plip_amd/csrc/libplipmi.so /tmp/lib_keep.so
for rep in 1 2; do for L in base early; do
  cp plip_amd/csrc/ab/lib_$L.so plip_amd/csrc/libplipmi.so
  echo "=== $L rep $rep" >> $O/ab.log
  for a in "2 12800 2304 768 0" "3 12800 3072 768 1"; do timeout 120 python tools/gpu_diag.py gemmtrace $a 2>&1 | grep -E "main loop" | cut -c1-200 >> $O/ab.log; done
  timeout 200 python tools/gpu_diag.py cold 2 3 2>&1 | grep -E "qkv|fc1" | cut -c1-200 >> $O/ab.log
  timeout 300 python bench.py --steps 20 --warmup 3 --overlap 0 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('one-stream', d['value'], d['ms_per_step'], d['windows']['ms_per_step'], [(k['name'][-28:], round(k['ms_per_step'],3)) for k in d['kernels'][:3]])" >> $O/ab.log
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('two-stream', d['value'], d['ms_per_step'], d['windows']['ms_per_step'])" >> $O/ab.log
done; done
cp plip_amd/csrc/ab/lib_early.so plip_amd/csrc/libplipmi.so
timeout 600 python -m pytest tests/test_gpu_gemm.py -m gpu -q -x 2>&1 | tail -2 >> $O/ab.log
cp /tmp/lib_keep.so plip_amd/csrc/libplipmi.so
cat $O/ab.log
