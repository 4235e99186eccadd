#!/bin/bash
# key-split attention with the second group's running max seeded from tile 0 (P operands = the unsplit kernel's): kernel tests, error statistics, split-point draws, timing
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; export TMPDIR=/tmp; out=gpurun_out; mkdir -p $out
timeout 600 python -m pytest tests/test_hip_kernels.py -k "attention" -m gpu -q -p no:cacheprovider 2>&1 | tail -8 > $out/r06al_pytest_attention_ks.log; cat $out/r06al_pytest_attention_ks.log
timeout 300 python tools/attn_ks_err.py 2>&1 | grep -v amdgpu.ids > $out/r06al_ks_err.log; cat $out/r06al_ks_err.log
timeout 900 python tools/ks_split_draws.py v1_vitl_518 vitl_518_t3600 2>&1 | grep -v "Warning\|warn\|amdgpu.ids" > $out/r06al_ks_split_draws.log; cat $out/r06al_ks_split_draws.log
(for c in "vitl b1 N3601" "vitl b1 N1370"; do KB_KS=1 timeout 120 ./tools/kbench attn "$c" 50; done) > $out/r06al_kbench_attn_ks.log 2>&1; cat $out/r06al_kbench_attn_ks.log
