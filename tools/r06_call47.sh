#!/bin/bash
# final .so (constant tidy in elementwise.hip): parity + kernel suites, smoke, driver-form bench
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; export TMPDIR=/tmp; out=gpurun_out; mkdir -p $out
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_hip_kernels.py tests/test_hip_v1.py -m gpu -q -p no:cacheprovider > $out/r06h_pytest_parity_kernels_v1.log 2>&1; tail -3 $out/r06h_pytest_parity_kernels_v1.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $out/r06h_smoke.log 2>&1; tail -1 $out/r06h_smoke.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/r06h_bench_driver_form.json 2> $out/r06h_bench_driver_form.err; head -c 330 $out/r06h_bench_driver_form.json; echo
