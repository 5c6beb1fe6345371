#!/bin/bash
# round 5, experiment 3: the fused text q/k/v + attention kernel -- bit-identity against the two kernels, the step with and without it
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r05_exp3; O=gpurun_out/r05_exp3; export TMPDIR=/tmp
t0=$(date +%s)
timeout 600 python -m pytest tests/test_gpu_attention.py -m gpu -q -x -k "fused or engine_runs" > $O/pytest_fused.log 2>&1; echo "pytest exit $? $(( $(date +%s) - t0 )) s" > $O/status.log
timeout 400 python tools/gpu_diag.py stepab f0 base f0 base > $O/stepab.log 2>&1
echo "stepab $(( $(date +%s) - t0 )) s" >> $O/status.log
timeout 300 python bench.py --steps 20 --warmup 3 --overlap 0 --no-cpu-baseline --no-extras > $O/bench_1s.json 2> $O/bench_1s.err
echo "bench1s $(( $(date +%s) - t0 )) s" >> $O/status.log
cat $O/status.log; tail -12 $O/pytest_fused.log; grep -E "one stream|max" $O/stepab.log
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_exp3/bench_1s.json').read().splitlines()[0])
print(d['value'], d['ms_per_step'])
for k in d['kernels'][:14]: print(k['name'][-60:], k.get('calls'), round(k['ms_per_step'],4))
PY
