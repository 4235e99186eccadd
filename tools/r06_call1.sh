#!/bin/bash
# round 6, GPU call 1: full GPU suite at HEAD (incl. the new large-input tests), tile-walk A/B in J/TFLOP (column groups of 8 vs 4), default bench
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; export TMPDIR=/tmp; out=gpurun_out; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -s > $out/r06a_pytest_gpu.log 2>&1; grep "passed\|failed\|error" $out/r06a_pytest_gpu.log | tail -3; grep "^FAILED\|^ERROR" $out/r06a_pytest_gpu.log | head -20
grep "^\[large\|^\[gate" $out/r06a_pytest_gpu.log > $out/r06a_gate_lines.log
for r in 1 2; do for d in 8 4; do
  echo "== PP_DBG (tile columns per group) = $d, round $r"
  KB_DBG=$d EN_SHAPES="qkv fc1" EN_NO_ATTN=1 bash tools/energy.sh r06a_dbg${d}_$r | grep "gemm"
done; done > $out/r06a_energy_colgroups.log 2>&1
cat $out/r06a_energy_colgroups.log
timeout 900 python bench.py > $out/r06a_bench_full.json 2> $out/r06a_bench_full.err; cat $out/r06a_bench_full.json | head -c 3000
