#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; export TMPDIR=/tmp; out=gpurun_out; mkdir -p $out
timeout 600 python -m pytest tests/test_hip_conv_ex.py -m gpu -q -p no:cacheprovider -x -k "ct3 or convt_conv3" > $out/r06e_pytest_ct3.log 2>&1; tail -40 $out/r06e_pytest_ct3.log
