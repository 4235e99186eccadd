#!/bin/bash
# round 6, GPU call 3: stream-count A/B of the production step (BATCH_SPLIT 2 / 3 / 4), two rounds, same box
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; export TMPDIR=/tmp; out=gpurun_out; mkdir -p $out
for r in 1 2; do for n in 2 3 4; do
  MOGE_BATCH_SPLIT=$n timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pcie --no-profile --no-power --no-autocast-pass 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('BATCH_SPLIT $n round $r: %.1f img/s  %.2f ms/step  p50 %.2f ms' % (d['value'], d['ms_per_step'], d['p50_latency_ms_batch1']))"
done; done > $out/r06c_ab_batch_split.log 2>&1
cat $out/r06c_ab_batch_split.log
