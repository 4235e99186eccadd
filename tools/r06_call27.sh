#!/bin/bash
# MODE 7 (W stream continuous across the tile boundary) against MODE 3: kbench interleaved + correctness, bit-identity tests of the GEMM suite, model A/B
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; export TMPDIR=/tmp; out=gpurun_out; mkdir -p $out
for f in qkv fc1 proj.h16 fc2.h16 outproj proj+fold fc2+fold; do KB_EXACT=1 KB_WX=1 KB_ROUNDS=3 timeout 200 ./tools/kbench gemm $f 10; done 2>&1 | grep -v "^   ts" > $out/r06w_kbench_gemm_wx.log; grep "interleaved\|FAIL\|bad [1-9]" $out/r06w_kbench_gemm_wx.log
timeout 900 python -m pytest tests/test_hip_gemm_pp.py -m gpu -q -p no:cacheprovider 2>&1 | tail -4
for r in 1 2; do for v in 0 1; do
  MOGE_PP_WX=$v timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pcie --no-autocast-pass 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); p = d.get('power') or {}; k = d['kernel_classes']
print('PP_WX $v round $r: %.1f img/s  %.2f ms/step  gemm_pp %.2f ms  p50 %.2f ms  %s W %s MHz' % (d['value'], d['ms_per_step'], k['gemm_pp']['ms_per_step'], d['p50_latency_ms_batch1'], p.get('avg_socket_w'), p.get('avg_sclk_mhz')))"
done; done > $out/r06w_ab_PP_WX.log 2>&1; cat $out/r06w_ab_PP_WX.log
