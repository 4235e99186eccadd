#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; export TMPDIR=/tmp; out=gpurun_out; mkdir -p $out
timeout 300 python tools/attn_ks_err.py > $out/r06aj_ks_err.log 2>&1; cat $out/r06aj_ks_err.log
for ks in 0 1; do echo "== MOGE_ATTN_KS=$ks"; MOGE_ATTN_KS=$ks timeout 600 python -m pytest tests/test_hip_v1.py -m gpu -q -p no:cacheprovider -s -k "fp16_mode and vitl" 2>&1 | grep "^\[gate\|passed\|failed\|AssertionError"; done > $out/r06aj_v1_ks01.log 2>&1; cat $out/r06aj_v1_ks01.log
