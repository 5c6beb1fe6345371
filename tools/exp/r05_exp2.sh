#!/bin/bash
# round 5, experiment 2: the whole gpu suite (no -x) on the round's host/ABI changes + folded plane re-coding, the bench line, small-batch tile A/B
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r05_exp2; O=gpurun_out/r05_exp2; export TMPDIR=/tmp
t0=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $? $(( $(date +%s) - t0 )) s" > $O/status.log
timeout 600 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err; echo "bench exit $? $(( $(date +%s) - t0 )) s" >> $O/status.log
for B in 32 64 128; do
  STEPAB_BATCH=$B timeout 300 python tools/gpu_diag.py stepab base "6>7" "6>7,2>7" "6>7,2>7,3>7" > $O/stepab_b$B.log 2>&1
done
echo "stepab $(( $(date +%s) - t0 )) s" >> $O/status.log
cat $O/status.log; tail -15 $O/pytest_gpu.log; for B in 32 64 128; do grep -E "one stream" $O/stepab_b$B.log; done
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05_exp2/bench.json'))
print({k:d.get(k) for k in ('value','ms_per_step','text_f16_layers','rccl_ranks','rank_devices','rccl_one_rank','logits_max_abs_err')})
print(d['roofline']); print(d.get('vitl14_336_b64'))
PY
