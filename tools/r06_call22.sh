#!/bin/bash
# side-stream tap LayerNorms (TAP_STREAM): parity tests that pin the taps, batch-1 A/B
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; export TMPDIR=/tmp; out=gpurun_out; mkdir -p $out
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_v1.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -4
AB_SUFFIX=_b1 BENCH_ARGS="--batch 1" AB_VAR=TAP_STREAM AB_VALS="0 1" bash tools/gpu_call.sh r06t ab
