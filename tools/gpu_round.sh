#!/bin/bash
# One gpurun call: diagnostics, the gpu test suite, the bench line and a rocprofv3 kernel trace.
# usage: gpurun --timeout 1500 -- 'bash tools/gpu_round.sh [sections...]'
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
SECTIONS="${@:-gemm tiny vitb32 gemmbench e2e pytest bench rocprof}"
echo "sections: $SECTIONS" > gpurun_out/status.log
for s in $SECTIONS; do
  t0=$(date +%s)
  case $s in
    pytest)  timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1 ;;
    pytestall) timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1 ;;
    pytestprints) timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -m gpu -q -s -k "vitl14 or heavy_tailed or config3 or full_matrix or text_tower_f16" 2>&1 | grep -E "cos|err|passed|failed" > gpurun_out/pytest_prints.log ;;
    bench)   timeout 600 python bench.py --steps 20 --warmup 3 > gpurun_out/bench.json 2> gpurun_out/bench.err ;;
    rocprof) (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof" -o plip -- python "$OLDPWD/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-profile --no-extras > "$OLDPWD/gpurun_out/rocprof_bench.json" 2> "$OLDPWD/gpurun_out/rocprof.err")
             (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof1s" -o plip -- python "$OLDPWD/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-profile --no-extras --overlap 0 > "$OLDPWD/gpurun_out/rocprof_bench_1stream.json" 2>> "$OLDPWD/gpurun_out/rocprof.err") ;;
    libprof) (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/proflib" -o lib -- python "$OLDPWD/tools/gpu_diag.py" libgemm > "$OLDPWD/gpurun_out/libprof.log" 2>&1) ;;
    bench1s) timeout 600 python bench.py --steps 20 --warmup 3 --overlap 0 --no-cpu-baseline --no-extras > gpurun_out/bench_1stream.json 2>> gpurun_out/bench.err ;;
    benchf16) timeout 600 python bench.py --steps 20 --warmup 3 --dtype f16 --no-cpu-baseline --no-extras > gpurun_out/bench_f16.json 2>> gpurun_out/bench.err ;;
    config3f16) timeout 900 python tools/config3_shard.py --dtype f16 > gpurun_out/config3_shard_f16.json 2> gpurun_out/config3_shard_f16.err ;;
    cold)    timeout 400 python tools/gpu_diag.py cold 2 3 4 5 6 > gpurun_out/diag_cold.log 2>&1 ;;
    pmc)     # hardware counters of the dominant GEMM (own passes, kernel-trace only -- see MI355X_MICROARCH rocprofv3 notes)
             for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES" \
                         "GRBM_GUI_ACTIVE FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_WAVES"; do
               tag=$(echo $pass | cut -d' ' -f1)
               (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d "$OLDPWD/gpurun_out/pmc_$tag" -o pmc -- python "$OLDPWD/tools/gpu_diag.py" gemmone ${PMC_ARGS:-6 12800 768 3072 2} >> "$OLDPWD/gpurun_out/pmc.log" 2>&1)
             done ;;
    trace)   echo "${TRACE_ARGS:-6 12800 768 3072 2;5 12800 768 3072 2;4 12800 768 3072 2;6 12800 768 768 2;6 19712 512 2048 2;6 19712 512 512 2;2 12800 2304 768 0;3 12800 3072 768 1;2 19712 1536 512 0;3 19712 2048 512 1}" | tr ';' '\n' | while read a; do
               timeout 120 python tools/gpu_diag.py gemmtrace $a >> gpurun_out/diag_gemmtrace.log 2>&1; done ;;
    # (the K-loop ablation of profiles/r03_gemm_kloop.txt needs the hooks of commit ba1f4b8: they cost 4-9 % of the K loop's
    #  cycles when compiled in, so the product kernels no longer carry them)
    pmcbench) rm -rf gpurun_out/pmc_bench_*
             for ov in 0 1; do for pass in "FETCH_SIZE" "WRITE_SIZE"; do
               (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d "$OLDPWD/gpurun_out/pmc_bench_${pass}_ov$ov" -o pmc -- python "$OLDPWD/bench.py" --steps 3 --warmup 1 --overlap $ov --no-cpu-baseline --no-profile --no-extras >> "$OLDPWD/gpurun_out/pmcbench.log" 2>&1)
             done; done
             python tools/pmc_summary.py gpurun_out/pmc_bench_FETCH_SIZE_ov0 gpurun_out/pmc_bench_WRITE_SIZE_ov0 gpurun_out/pmc_bench_FETCH_SIZE_ov1 gpurun_out/pmc_bench_WRITE_SIZE_ov1 > gpurun_out/pmc_traffic.json 2>> gpurun_out/pmcbench.log ;;
    torchrun1) timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-fp32-tower > gpurun_out/bench_torchrun1.json 2> gpurun_out/bench_torchrun1.err ;;
    pmcstep) # hardware counters of EVERY kernel of the step (one stream), separate passes; then HBM traffic in both stream modes
             rm -rf gpurun_out/pmcstep_* gpurun_out/pmc_bench_*
             i=0
             for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES" \
                         "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS SQ_INSTS_VALU SQ_LDS_IDX_ACTIVE"; do
               i=$((i+1))
               (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d "$OLDPWD/gpurun_out/pmcstep_$i" -o pmc -- python "$OLDPWD/bench.py" --steps 3 --warmup 1 --overlap 0 --no-cpu-baseline --no-profile --no-extras >> "$OLDPWD/gpurun_out/pmcstep.log" 2>&1)
             done
             for ov in 0 1; do for pass in "FETCH_SIZE" "WRITE_SIZE"; do
               (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d "$OLDPWD/gpurun_out/pmc_bench_${pass}_ov$ov" -o pmc -- python "$OLDPWD/bench.py" --steps 3 --warmup 1 --overlap $ov --no-cpu-baseline --no-profile --no-extras >> "$OLDPWD/gpurun_out/pmcstep.log" 2>&1)
             done; done
             python tools/pmc_table.py gpurun_out/pmcstep_1 gpurun_out/pmcstep_2 gpurun_out/pmc_bench_FETCH_SIZE_ov0 gpurun_out/pmc_bench_WRITE_SIZE_ov0 > gpurun_out/pmc_step_counters.txt 2>> gpurun_out/pmcstep.log
             python tools/pmc_summary.py gpurun_out/pmc_bench_FETCH_SIZE_ov0 gpurun_out/pmc_bench_WRITE_SIZE_ov0 gpurun_out/pmc_bench_FETCH_SIZE_ov1 gpurun_out/pmc_bench_WRITE_SIZE_ov1 > gpurun_out/pmc_traffic.json 2>> gpurun_out/pmcstep.log ;;
    config3) timeout 900 python tools/config3_shard.py > gpurun_out/config3_shard.json 2> gpurun_out/config3_shard.err ;;
    smoke)   timeout 600 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1 ;;
    *)       timeout 600 python tools/gpu_diag.py $s > gpurun_out/diag_$s.log 2>&1 ;;
  esac
  echo "$s exit $? ($(( $(date +%s) - t0 )) s)" >> gpurun_out/status.log
done
cat gpurun_out/status.log
