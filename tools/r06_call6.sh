#!/bin/bash
# round 6, GPU call 6: model-level parity with the fused ConvTranspose2d + 3x3 (CT3) on, then bench A/B FUSE_CT3 = 0 / 1 on the same box
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; export TMPDIR=/tmp; out=gpurun_out; mkdir -p $out
timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_hip_conv_ex.py tests/test_hip_host_example.py -m gpu -q -p no:cacheprovider -s > $out/r06f_pytest_parity_ct3.log 2>&1; grep "passed\|failed\|error" $out/r06f_pytest_parity_ct3.log | tail -3; grep "^FAILED\|^ERROR" $out/r06f_pytest_parity_ct3.log | head
grep "^\[gate" $out/r06f_pytest_parity_ct3.log > $out/r06f_gate_lines.log
for r in 1 2; do for v in 0 1; do
  MOGE_FUSE_CT3=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pcie --no-power --no-autocast-pass 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); k = d['kernel_classes']
print('FUSE_CT3 $v round $r: %.1f img/s  %.2f ms/step  p50 %.2f ms | conv %.2f ms (%.0f TF/s algorithmic)  gemm_pp %.2f  attn %.2f' % (d['value'], d['ms_per_step'], d['p50_latency_ms_batch1'], k['conv']['ms_per_step'], k['conv']['tflops'], k['gemm_pp']['ms_per_step'], k['attn']['ms_per_step']))"
done; done > $out/r06f_ab_ct3.log 2>&1
cat $out/r06f_ab_ct3.log
