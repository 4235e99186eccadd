#!/bin/bash
# one image at the final tree, launch by launch (key-split attention, 181 launches)
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; export TMPDIR=/tmp; out=gpurun_out; mkdir -p $out
rm -rf /tmp/lt; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/lt -o lt -- python bench.py --batch 1 --steps 40 --warmup 5 --no-cpu-baseline --no-pcie --no-profile --no-power --no-autocast-pass --no-latency > /dev/null 2>&1
python3 tools/trace_b1_steps.py /tmp/lt/lt_kernel_trace.csv > $out/r06g_b1_step_anatomy.log 2>&1; head -14 $out/r06g_b1_step_anatomy.log | cut -c1-140; tail -3 $out/r06g_b1_step_anatomy.log
python3 tools/trace_summary.py /tmp/lt/lt_kernel_trace.csv 40 > $out/r06g_b1_kernels_by_grid.csv; head -12 $out/r06g_b1_kernels_by_grid.csv
for r in 1 2; do timeout 300 python bench.py --batch 1 --steps 40 --warmup 5 --no-cpu-baseline --no-pcie --no-power --no-autocast-pass 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('B=1: %.1f img/s %.3f ms/step p50 %.3f' % (d['value'], d['ms_per_step'], d['p50_latency_ms_batch1']))"; done > $out/r06g_bench_b1.log; cat $out/r06g_bench_b1.log
