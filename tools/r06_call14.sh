#!/bin/bash
# stream-K attention (write-through slots instead of device-scope fences) + fused LN finalize: kbench, batch-1 A/B, the GPU suite
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; export TMPDIR=/tmp; out=gpurun_out; mkdir -p $out
for f in "vitl b1 N3601" "vitb b1" "vitl b1 N1370"; do KB_SK=1 timeout 120 ./tools/kbench attn "$f" 30; done > $out/r06m_kbench_attn_sk.log 2>&1; cat $out/r06m_kbench_attn_sk.log
AB_SUFFIX=_b1 BENCH_ARGS="--batch 1" AB_VAR=ATTN_SK AB_VALS="0 1" bash tools/gpu_call.sh r06m ab
AB_SUFFIX=_b1 BENCH_ARGS="--batch 1" AB_VAR=LN_FINALIZE_FUSED AB_VALS="0 1" bash tools/gpu_call.sh r06m ab
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $out/r06m_pytest_gpu.log 2>&1; tail -40 $out/r06m_pytest_gpu.log
