#!/bin/bash
# spread of the small configs: vits-normal / vitb-normal at batch 8, three runs each on one box
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; export TMPDIR=/tmp; out=gpurun_out; mkdir -p $out
for r in 1 2 3; do for cfg in "--config moge-2-vits-normal --batch 8" "--config moge-2-vitb-normal --batch 8"; do
  timeout 300 python bench.py $cfg --steps 20 --warmup 5 --no-cpu-baseline --no-pcie --no-power --no-autocast-pass 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); k = d['kernel_classes']
print('$cfg round $r: %.1f img/s %.2f ms/step p50 %.2f | conv %.2f gemm_pp %.2f gemm %.2f attn %.2f' % (d['value'], d['ms_per_step'], d['p50_latency_ms_batch1'], k['conv']['ms_per_step'], k['gemm_pp']['ms_per_step'], k['gemm']['ms_per_step'], k['attn']['ms_per_step']))"
done; done > $out/r06v_small_configs_spread.log 2>&1; cat $out/r06v_small_configs_spread.log
