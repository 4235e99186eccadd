#!/bin/bash
# Same-box A/B of the whole bench step between two settings of one MOGE_<KEY> switch (clocks differ box to box: only this comparison is valid).
#   tools/ab_model.sh KEY A B [rounds]      -> gpurun_out/ab_<KEY>.log   (images/s, ms/step and the kernel-class table per run)
key=$1; a=$2; b=$3; rounds=${4:-3}
out=gpurun_out/ab_$key.log; mkdir -p gpurun_out; : > $out
for r in $(seq $rounds); do
  for v in $a $b; do
    env MOGE_$key=$v python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pcie 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
kc = d.get('kernel_classes', {})
print('$key=$v', 'img/s %.2f' % d['value'], 'ms/step %.2f' % d['ms_per_step'], 'b1 %.3f ms' % d.get('p50_latency_ms_batch1', 0), ' '.join('%s %.2f' % (k, v['ms_per_step']) for k, v in kc.items()))
" >> $out
  done
done
cat $out
