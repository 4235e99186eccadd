#!/bin/bash
# batch-1 anatomy at HEAD: bench line, rocprofv3 kernel trace of the batch-1 steps, timeline (kernels in flight, idle gaps), latency-regime kbench
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; export TMPDIR=/tmp; out=gpurun_out; mkdir -p $out
tag=r06j
timeout 300 python bench.py --batch 1 --steps 30 --warmup 5 --no-cpu-baseline --no-pcie --no-power --no-autocast-pass > $out/${tag}_bench_b1.json 2>/dev/null
python -c "
import json
d = json.loads(open('$out/${tag}_bench_b1.json').read().strip().splitlines()[-1]); k = d['kernel_classes']
print('B=1: %.1f img/s %.3f ms/step p50 %.3f | ' % (d['value'], d['ms_per_step'], d['p50_latency_ms_batch1']) + ' '.join('%s %.3f' % (n, v['ms_per_step']) for n, v in k.items()))"
rm -rf /tmp/lt; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/lt -o lt -- python bench.py --batch 1 --steps 30 --warmup 5 --no-cpu-baseline --no-pcie --no-profile --no-power --no-autocast-pass --no-latency > /dev/null 2>&1
python3 tools/trace_summary.py /tmp/lt/lt_kernel_trace.csv 80 > $out/${tag}_b1_kernels_by_grid.csv
python3 tools/trace_overlap.py /tmp/lt/lt_kernel_trace.csv b1 > $out/${tag}_b1_timeline.log 2>&1; cat $out/${tag}_b1_timeline.log
python3 tools/trace_b1_steps.py /tmp/lt/lt_kernel_trace.csv > $out/${tag}_b1_step_anatomy.log 2>&1; tail -120 $out/${tag}_b1_step_anatomy.log
for f in b1.; do KB_LAT=1 KB_ROUNDS=2 timeout 200 ./tools/kbench gemm $f 20; done 2>&1 | grep -v "^   ts" > $out/${tag}_kbench_gemm_latency.log; grep interleaved $out/${tag}_kbench_gemm_latency.log
timeout 100 ./tools/kbench attn "b1 N" 20 > $out/${tag}_kbench_attn_b1.log 2>&1; cat $out/${tag}_kbench_attn_b1.log
