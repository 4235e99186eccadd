#!/bin/bash
# pipelined heads (HEAD_PIPE): bit-identity test, batch-1 A/B, anatomy
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; export TMPDIR=/tmp; out=gpurun_out; mkdir -p $out
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -p no:cacheprovider -k "head_streams or batch_split or error_behaviour" 2>&1 | tail -4
AB_SUFFIX=_b1 BENCH_ARGS="--batch 1" AB_VAR=HEAD_PIPE AB_VALS="0 1" bash tools/gpu_call.sh r06s ab
AB_SUFFIX=_b1_vitln BENCH_ARGS="--batch 1 --config moge-2-vitl-normal" AB_VAR=HEAD_PIPE AB_VALS="0 1" bash tools/gpu_call.sh r06s ab
rm -rf /tmp/lt; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/lt -o lt -- python bench.py --batch 1 --steps 30 --warmup 5 --no-cpu-baseline --no-pcie --no-profile --no-power --no-autocast-pass --no-latency > /dev/null 2>&1
python3 tools/trace_b1_steps.py /tmp/lt/lt_kernel_trace.csv > $out/r06s_b1_step_anatomy.log 2>&1; head -2 $out/r06s_b1_step_anatomy.log; tail -75 $out/r06s_b1_step_anatomy.log
