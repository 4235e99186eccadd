#!/bin/bash
# driver-form bench (--steps 20 --warmup 5), fused ConvTranspose2d + 3x3 off / on, three alternations on one box
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; export TMPDIR=/tmp; out=gpurun_out; mkdir -p $out
for r in 1 2 3; do for v in 0 1; do
  MOGE_FUSE_CT3=$v timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pcie --no-autocast-pass --no-profile 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); p = d.get('power') or {}
print('FUSE_CT3 $v round $r: %.1f img/s  %.2f ms/step  p50 %.2f ms  %s W %s MHz' % (d['value'], d['ms_per_step'], d['p50_latency_ms_batch1'], p.get('avg_socket_w'), p.get('avg_sclk_mhz')))"
done; done > $out/r06i_ab_ct3_driver_form.log 2>&1
cat $out/r06i_ab_ct3_driver_form.log
