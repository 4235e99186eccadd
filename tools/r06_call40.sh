#!/bin/bash
# four launches fewer in front of the first block (K padding + counters zeroed by preprocess_kernel, cls row written by the patch-embed epilogue): parity suites, old / new library at one image
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; export TMPDIR=/tmp; out=gpurun_out; mkdir -p $out
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_hip_v1.py tests/test_hip_kernels.py -m gpu -q -p no:cacheprovider -x > $out/r06ai_pytest_parity.log 2>&1; tail -4 $out/r06ai_pytest_parity.log
cp moge_amd/lib/libmoge_hip.so /tmp/new_lib.so
for r in 1 2 3; do for v in old new; do
  if [ $v = old ]; then cp tools/ab_old/libmoge_hip.so moge_amd/lib/libmoge_hip.so; else cp /tmp/new_lib.so moge_amd/lib/libmoge_hip.so; fi
  timeout 300 python bench.py --batch 1 --steps 40 --warmup 5 --no-cpu-baseline --no-pcie --no-power --no-autocast-pass 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('lib=$v', 'img/s %.2f' % d['value'], 'ms/step %.3f' % d['ms_per_step'], 'p50 %.3f ms' % d.get('p50_latency_ms_batch1', 0))"
done; done > $out/r06ai_ab_lib_b1.log 2>&1
cp /tmp/new_lib.so moge_amd/lib/libmoge_hip.so; cat $out/r06ai_ab_lib_b1.log
