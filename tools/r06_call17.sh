#!/bin/bash
# moge_sync: polled pinned status word vs blocking wait (SYNC_SPIN_US 0 / 20000) at one image; error path; batch-32 sanity
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; export TMPDIR=/tmp; out=gpurun_out; mkdir -p $out
AB_SUFFIX=_b1 BENCH_ARGS="--batch 1" AB_VAR=SYNC_SPIN_US AB_VALS="0 20000" bash tools/gpu_call.sh r06p ab
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -p no:cacheprovider -k "finite" 2>&1 | tail -4
for b in 2 4 8; do timeout 300 python bench.py --batch $b --steps 10 --warmup 3 --no-cpu-baseline --no-pcie --no-power --no-autocast-pass --no-latency --no-profile 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('batch $b: %.1f img/s %.3f ms/step' % (d['value'], d['ms_per_step']))"; done
