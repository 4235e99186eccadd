#!/bin/bash
# split-K emulation for the batch-1 fc2: same FLOPs as (M 3601, K 4096) with 2x / 4x the rows and K / 2, K / 4
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; export TMPDIR=/tmp; out=gpurun_out; mkdir -p $out
for f in b1.fc2 b1.fc2s2 b4.proj b1.proj; do KB_EXACT=1 KB_LAT=1 KB_ROUNDS=2 timeout 200 ./tools/kbench gemm $f 20; done 2>&1 | grep -v "^   ts" > $out/r06k_kbench_splitk_emulation.log; grep interleaved $out/r06k_kbench_splitk_emulation.log
