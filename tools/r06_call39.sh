#!/bin/bash
# key-split attention as the default for single images: the re-scoped model-level tests
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; export TMPDIR=/tmp; out=gpurun_out; mkdir -p $out
timeout 1200 python -m pytest tests/test_hip_parity.py -m gpu -q -p no:cacheprovider -s -k "throughput_kernels or batch_of_32 or properties_at_baseline or key_split or within_reference_fp16 or fp16_mode" > $out/r06ah_pytest_ks_model.log 2>&1; grep "passed\|failed\|error" $out/r06ah_pytest_ks_model.log | tail -3; grep "^FAILED\|^ERROR\|Error\|assert" $out/r06ah_pytest_ks_model.log | head -20
grep "^\[gate" $out/r06ah_pytest_ks_model.log > $out/r06ah_gate_lines.log; wc -l $out/r06ah_gate_lines.log
