#!/bin/bash
# layernorm_kernel with one row per wave for few rows: kernel + parity tests, its duration in a batch-1 trace
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; export TMPDIR=/tmp; out=gpurun_out; mkdir -p $out
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_hip_parity.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3
rm -rf /tmp/lt; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/lt -o lt -- python bench.py --batch 1 --steps 30 --warmup 5 --no-cpu-baseline --no-pcie --no-profile --no-power --no-autocast-pass --no-latency > /dev/null 2>&1
python3 tools/trace_summary.py /tmp/lt/lt_kernel_trace.csv 80 > $out/r06af_b1_kernels_by_grid.csv; grep "layernorm\|recover\|total" $out/r06af_b1_kernels_by_grid.csv
python3 tools/trace_b1_steps.py /tmp/lt/lt_kernel_trace.csv > $out/r06af_b1_step_anatomy.log 2>&1; head -1 $out/r06af_b1_step_anatomy.log; tail -1 $out/r06af_b1_step_anatomy.log
timeout 300 python bench.py --batch 1 --steps 30 --warmup 5 --no-cpu-baseline --no-pcie --no-power --no-autocast-pass 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('B=1: %.1f img/s %.3f ms/step p50 %.3f' % (d['value'], d['ms_per_step'], d['p50_latency_ms_batch1']))"
