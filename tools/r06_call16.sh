#!/bin/bash
# batch 1: moge_sync with one host wait (p50), FUSE_CT3 off / on at one image, error path of the status word; experiments build: stream-K tests
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; export TMPDIR=/tmp; out=gpurun_out; mkdir -p $out
AB_SUFFIX=_b1 BENCH_ARGS="--batch 1" AB_VAR=FUSE_CT3 AB_VALS="0 1" bash tools/gpu_call.sh r06o ab
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -p no:cacheprovider -k "finite or overflow or determin or half_model" 2>&1 | tail -4
rm -rf /tmp/lt; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/lt -o lt -- python bench.py --batch 1 --steps 30 --warmup 5 --no-cpu-baseline --no-pcie --no-profile --no-power --no-autocast-pass --no-latency > /dev/null 2>&1
python3 tools/trace_b1_steps.py /tmp/lt/lt_kernel_trace.csv > $out/r06o_b1_step_anatomy.log 2>&1; head -2 $out/r06o_b1_step_anatomy.log; tail -8 $out/r06o_b1_step_anatomy.log
