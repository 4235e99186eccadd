#!/bin/bash
# Same-box A/B of the whole bench step between two BUILDS of the library: tools/_ab/old/libmoge_hip.so (built from another commit by hand)
# against the in-tree one.  Runs on the GPU box's scratch copy of the repo (the in-tree file is swapped there, nothing is committed).
#   tools/ab_lib.sh [rounds]      -> gpurun_out/ab_lib.log
rounds=${1:-3}
out=gpurun_out/ab_lib.log; mkdir -p gpurun_out; : > $out
cp moge_amd/lib/libmoge_hip.so /tmp/new_lib.so
for r in $(seq $rounds); do
  for v in old new; do
    if [ $v = old ]; then cp tools/_ab/old/libmoge_hip.so moge_amd/lib/libmoge_hip.so; else cp /tmp/new_lib.so moge_amd/lib/libmoge_hip.so; fi
    python bench.py $AB_ARGS --steps 6 --warmup 2 --no-cpu-baseline --no-pcie 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
kc = d.get('kernel_classes', {})
print('lib=$v', 'img/s %.2f' % d['value'], 'ms/step %.2f' % d['ms_per_step'], 'b1 %.3f ms' % d.get('p50_latency_ms_batch1', 0), ' '.join('%s %.2f' % (k, v['ms_per_step']) for k, v in kc.items()))
" >> $out
  done
done
cp /tmp/new_lib.so moge_amd/lib/libmoge_hip.so
cat $out
