#!/bin/bash
# round-4 experiment 2: what the vendor library runs on our shapes (kernel names = its tile choices), and how power-limited the step is
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/exp2; mkdir -p $O
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$O/proflib" -o lib -- python "$OLDPWD/tools/gpu_diag.py" libgemm 2 3 6 > "$OLDPWD/$O/libprof.log" 2>&1); echo "libprof rc $?" >> $O/status.log
# power / clock telemetry while the step loops
( for i in $(seq 1 40); do rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|fclk|Temperature \(Sensor (junction|edge)" | tr '\n' ';'; echo; sleep 0.25; done > $O/smi_during.log ) &
SMI=$!
timeout 120 python tools/gpu_diag.py power > $O/power.log 2>&1; echo "power rc $?" >> $O/status.log
wait $SMI
rocm-smi --showpower --showclocks --showmaxpower 2>/dev/null > $O/smi_idle.log
rocm-smi -a 2>/dev/null | head -150 > $O/smi_all.log
cat $O/status.log; cat $O/power.log | tail -8; head -5 $O/smi_during.log; sed -n 15,25p $O/smi_during.log
