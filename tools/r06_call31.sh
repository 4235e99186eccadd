#!/bin/bash
# the escape hatches of this round still pass the parity suite: HEAD_PIPE 0 + LN_FINALIZE_FUSED 0
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; export TMPDIR=/tmp; out=gpurun_out; mkdir -p $out
MOGE_HEAD_PIPE=0 MOGE_LN_FINALIZE_FUSED=0 timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_hip_v1.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3 > $out/r06ab_pytest_escape_hatches.log; cat $out/r06ab_pytest_escape_hatches.log
