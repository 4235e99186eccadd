#!/bin/bash
# round 6, GPU call 4 (library built with --experiments): the occupancy-2 GEMM question on the kernels that exist -
# x:pp64-2wg = 256 x 128 tiles, K-tile 32 (64-byte rows), 72 KiB LDS, TWO workgroups per CU, against the product kernel, same box
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; export TMPDIR=/tmp; out=gpurun_out; mkdir -p $out
for s in qkv fc1 proj.h16 fc2.h16; do
  KB_EXACT=1 KB_EXP=1 KB_ROUNDS=2 timeout 300 ./tools/kbench gemm $s 10
done > $out/r06d_kbench_gemm_occ2.log 2>&1
grep -v "^ " $out/r06d_kbench_gemm_occ2.log | head -80
