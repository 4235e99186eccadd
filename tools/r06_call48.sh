#!/bin/bash
# the round's last tree: full GPU suite, gate lines, smoke, default bench, driver-form bench
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; export TMPDIR=/tmp; out=gpurun_out; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -s > $out/r06i_pytest_gpu.log 2>&1; grep "passed\|failed\|error" $out/r06i_pytest_gpu.log | tail -3; grep "^FAILED\|^ERROR" $out/r06i_pytest_gpu.log | head
grep "^\[large\|^\[gate" $out/r06i_pytest_gpu.log > $out/r06i_gate_lines.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $out/r06i_smoke.log 2>&1; tail -1 $out/r06i_smoke.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/r06i_bench_driver_form.json 2> $out/r06i_bench_driver_form.err; head -c 300 $out/r06i_bench_driver_form.json; echo
timeout 900 python bench.py > $out/r06i_bench_full.json 2> $out/r06i_bench_full.err; head -c 300 $out/r06i_bench_full.json; echo
