#!/bin/bash
# stream-K attention: kernel tests, kbench at batch 1 / 2, model A/B at batch 1
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; export TMPDIR=/tmp; out=gpurun_out; mkdir -p $out
timeout 900 python -m pytest tests/test_hip_kernels.py -k "attention" -m gpu -q -p no:cacheprovider -x > $out/r06l_pytest_attention_sk.log 2>&1; tail -15 $out/r06l_pytest_attention_sk.log
for f in "vitl b1 N3601" "vitb b1" "vitl b1 N1370" "vitl b2"; do KB_SK=1 timeout 120 ./tools/kbench attn "$f" 30; done > $out/r06l_kbench_attn_sk.log 2>&1; cat $out/r06l_kbench_attn_sk.log
AB_SUFFIX=_b1 BENCH_ARGS="--batch 1" AB_VAR=ATTN_SK AB_VALS="0 1" bash tools/gpu_call.sh r06l ab
