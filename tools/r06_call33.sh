#!/bin/bash
# ViT-S (N = 384: its proj / fc2 run on gemm_glds_kernel at every batch size): fused LN finalize off / on at batch 8 and batch 32
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; export TMPDIR=/tmp; out=gpurun_out; mkdir -p $out
AB_SUFFIX=_vits8 BENCH_ARGS="--config moge-2-vits-normal --batch 8" AB_VAR=LN_FINALIZE_FUSED AB_VALS="0 1" bash tools/gpu_call.sh r06ad ab
AB_SUFFIX=_vits32 BENCH_ARGS="--config moge-2-vits-normal --batch 32" AB_VAR=LN_FINALIZE_FUSED AB_VALS="0 1" bash tools/gpu_call.sh r06ad ab
