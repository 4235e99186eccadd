#!/bin/bash
# rocprofv3 trace + PMC passes at the final tree (gemm_pp.hip gained two experiment-only stamps: same product ISA, new source hash)
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; export TMPDIR=/tmp; out=gpurun_out; mkdir -p $out
PROF_TIMEOUT=300 bash tools/profile_round.sh r06x > $out/r06x_profile_round.log 2>&1; tail -3 $out/r06x_profile_round.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/r06x_bench_driver_form.json 2> $out/r06x_bench_driver_form.err; head -c 300 $out/r06x_bench_driver_form.json; echo
