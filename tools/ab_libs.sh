# A/B of several builds of libplipmi.so on ONE box: usage  bash tools/ab_libs.sh <tag>...   (plip_amd/csrc/ab/lib_<tag>.so)
mkdir -p gpurun_out; : > gpurun_out/ab.log
cp plip_amd/csrc/libplipmi.so /tmp/lib_keep.so
for rep in 1 2; do for L in "$@"; do
  cp plip_amd/csrc/ab/lib_$L.so plip_amd/csrc/libplipmi.so
  echo "=== $L rep $rep" >> gpurun_out/ab.log
  [ -n "$AB_TRACE" ] && for a in "6 12800 768 3072 2" "2 12800 2304 768 0" "3 12800 3072 768 1"; do timeout 120 python tools/gpu_diag.py gemmtrace $a 2>&1 | grep -E "main loop" >> gpurun_out/ab.log; done
  timeout 300 python bench.py --steps 20 --warmup 3 --overlap 0 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('one-stream', d['value'], d['ms_per_step'], d['windows']['ms_per_step'], [(k['name'][-28:], round(k['ms_per_step'],3)) for k in d['kernels'][:4]])" >> gpurun_out/ab.log
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('two-stream', d['value'], d['ms_per_step'], d['windows']['ms_per_step'])" >> gpurun_out/ab.log
done; done
cp /tmp/lib_keep.so plip_amd/csrc/libplipmi.so
cat gpurun_out/ab.log
