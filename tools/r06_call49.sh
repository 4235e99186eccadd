#!/bin/bash
# --experiments build at the round's last tree: the kernels that share attn_pp16mq's body (stream-K, ping-pong, single-stream) and the other experiment-only tests
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; export TMPDIR=/tmp; out=gpurun_out; mkdir -p $out
timeout 1200 python -m pytest tests/test_hip_kernels.py tests/test_hip_gemm_pp.py tests/test_hip_conv_ex.py -m gpu -q -p no:cacheprovider > $out/r06j_pytest_experiments_build.log 2>&1; tail -4 $out/r06j_pytest_experiments_build.log
