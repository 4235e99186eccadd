#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; export TMPDIR=/tmp; out=gpurun_out; mkdir -p $out
(timeout 600 python tools/ks_forward_sensitivity.py v1_vitl_518 2500; timeout 600 python tools/ks_forward_sensitivity.py v1_vitl_train_config_518 2500) 2>&1 | grep -v "Warning\|warn\|amdgpu.ids" > $out/r06am_ks_forward_sensitivity.log; cat $out/r06am_ks_forward_sensitivity.log
