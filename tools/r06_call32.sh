#!/bin/bash
# one image: attention QB = 2 limited to two workgroups per CU (no CU takes a third one of a grid that fits two per CU) against the default three
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; export TMPDIR=/tmp; out=gpurun_out; mkdir -p $out
AB_SUFFIX=_b1 BENCH_ARGS="--batch 1" AB_VAR=ATTN_Q2_PER_CU AB_VALS="3 2" bash tools/gpu_call.sh r06ac ab
AB_SUFFIX=_b1_vitb BENCH_ARGS="--batch 1 --config moge-2-vitb-normal" AB_VAR=ATTN_Q2_PER_CU AB_VALS="3 2" bash tools/gpu_call.sh r06ac ab
