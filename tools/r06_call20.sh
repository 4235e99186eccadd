#!/bin/bash
# batch 1: recover_kernel with 512 / 1024 threads
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; export TMPDIR=/tmp; out=gpurun_out; mkdir -p $out
AB_SUFFIX=_b1 BENCH_ARGS="--batch 1" AB_VAR=REC_THREADS AB_VALS="1024 512" bash tools/gpu_call.sh r06r ab
