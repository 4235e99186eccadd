#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; export TMPDIR=/tmp; out=gpurun_out; mkdir -p $out
timeout 900 python tools/ks_split_draws.py v1_vitl_518 v1_vitl_train_config_518 vitl_518_t3600 > $out/r06ak_ks_split_draws.log 2>&1; grep -v Warning $out/r06ak_ks_split_draws.log | tail -70
