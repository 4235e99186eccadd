#!/bin/bash
# fused LN finalize (latency-regime GEMMs), scale head behind the heads, grouped loads in ct3_border: batch-1 A/B + anatomy, GPU suite, batch-32 bench
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; export TMPDIR=/tmp; out=gpurun_out; mkdir -p $out
AB_SUFFIX=_b1 BENCH_ARGS="--batch 1" AB_VAR=LN_FINALIZE_FUSED AB_VALS="0 1" bash tools/gpu_call.sh r06n ab
rm -rf /tmp/lt; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/lt -o lt -- python bench.py --batch 1 --steps 30 --warmup 5 --no-cpu-baseline --no-pcie --no-profile --no-power --no-autocast-pass --no-latency > /dev/null 2>&1
python3 tools/trace_summary.py /tmp/lt/lt_kernel_trace.csv 80 > $out/r06n_b1_kernels_by_grid.csv
python3 tools/trace_b1_steps.py /tmp/lt/lt_kernel_trace.csv > $out/r06n_b1_step_anatomy.log 2>&1; head -3 $out/r06n_b1_step_anatomy.log; tail -45 $out/r06n_b1_step_anatomy.log
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x > $out/r06n_pytest_gpu.log 2>&1; tail -15 $out/r06n_pytest_gpu.log
bash tools/gpu_call.sh r06n bench
