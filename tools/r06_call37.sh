#!/bin/bash
# ct3_border_kernel with the K-steps split over a workgroup's four waves: kernel + parity tests, durations at batch 1 and batch 32
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; export TMPDIR=/tmp; out=gpurun_out; mkdir -p $out
timeout 900 python -m pytest tests/test_hip_conv_ex.py tests/test_hip_parity.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3
rm -rf /tmp/lt; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/lt -o lt -- python bench.py --batch 1 --steps 30 --warmup 5 --no-cpu-baseline --no-pcie --no-profile --no-power --no-autocast-pass --no-latency > /dev/null 2>&1
python3 tools/trace_summary.py /tmp/lt/lt_kernel_trace.csv 80 | grep "ct3_border\|total"
rm -rf /tmp/lt; MOGE_BATCH_SPLIT=0 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/lt -o lt -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pcie --no-profile --no-power --no-autocast-pass --no-latency > /dev/null 2>&1
python3 tools/trace_summary.py /tmp/lt/lt_kernel_trace.csv 80 | grep "ct3_border\|total"
timeout 300 python bench.py --batch 1 --steps 30 --warmup 5 --no-cpu-baseline --no-pcie --no-power --no-autocast-pass 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('B=1: %.1f img/s %.3f ms/step p50 %.3f' % (d['value'], d['ms_per_step'], d['p50_latency_ms_batch1']))"
