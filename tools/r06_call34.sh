#!/bin/bash
# recover_kernel with 1024 / 512 / 256 threads at one image: kernel duration from a rocprofv3 trace + p50
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; export TMPDIR=/tmp; out=gpurun_out; mkdir -p $out
for t in 1024 512 256; do
  rm -rf /tmp/lt; MOGE_REC_THREADS=$t timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/lt -o lt -- python bench.py --batch 1 --steps 30 --warmup 5 --no-cpu-baseline --no-pcie --no-profile --no-power --no-autocast-pass --no-latency > /dev/null 2>&1
  python3 tools/trace_summary.py /tmp/lt/lt_kernel_trace.csv 80 | grep "recover_kernel" | sed "s/^/REC_THREADS $t: /"
done > $out/r06ae_recover_threads.log 2>&1; cat $out/r06ae_recover_threads.log
AB_SUFFIX=_b1 BENCH_ARGS="--batch 1" AB_VAR=REC_THREADS AB_VALS="1024 512 256" bash tools/gpu_call.sh r06ae ab
