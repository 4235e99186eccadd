#!/bin/bash
# attn_pp16ks_kernel (key range split inside an 8-wave workgroup): kernel tests, kbench at the one-image shapes, model A/B at one image
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; export TMPDIR=/tmp; out=gpurun_out; mkdir -p $out
timeout 600 python -m pytest tests/test_hip_kernels.py -k "attention" -m gpu -q -p no:cacheprovider 2>&1 | tail -8 > $out/r06ag_pytest_attention_ks.log; cat $out/r06ag_pytest_attention_ks.log
(for c in "vitl b1 N3601" "vitb b1 N3601" "vitl b1 N1370" "vitl b2 N3601"; do KB_KS=1 timeout 120 ./tools/kbench attn "$c" 50; done) > $out/r06ag_kbench_attn_ks.log 2>&1; cat $out/r06ag_kbench_attn_ks.log
AB_SUFFIX=_b1 BENCH_ARGS="--batch 1" AB_VAR=ATTN_KS AB_VALS="0 1" bash tools/gpu_call.sh r06ag ab
