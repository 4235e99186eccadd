#!/bin/bash
# MODE 7 against MODE 3 SUSTAINED (1500-launch samples: the chip settles at its power-limited clock), interleaved
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; export TMPDIR=/tmp; out=gpurun_out; mkdir -p $out
for f in qkv fc1 proj.h16 fc2.h16; do KB_EXACT=1 KB_WX=1 KB_ROUNDS=3 timeout 300 ./tools/kbench gemm $f 1500; done 2>&1 | grep -v "^   ts" > $out/r06w_kbench_gemm_wx_sustained.log; grep "interleaved\|FAIL\|bad [1-9]" $out/r06w_kbench_gemm_wx_sustained.log
for r in 1 2 3; do for v in 0 1; do
  MOGE_PP_WX=$v timeout 300 python bench.py --gpus 1 --steps 30 --warmup 5 --no-cpu-baseline --no-pcie --no-autocast-pass --no-latency --no-power --no-profile 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('PP_WX $v round $r: %.1f img/s  %.2f ms/step' % (d['value'], d['ms_per_step']))"
done; done > $out/r06w_ab_PP_WX_30steps.log 2>&1; cat $out/r06w_ab_PP_WX_30steps.log
