#!/bin/bash
# attention sustained (600-launch samples = ~1 s at the power-limited clock): row sums on the VALU (pp16: 32 queries per wave, 3 waves per SIMD) against row sums on the matrix pipe (mq<2>, mq<4>)
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; export TMPDIR=/tmp; out=gpurun_out; mkdir -p $out
timeout 300 ./tools/kbench attn "vitl b32 N3601" 600 > $out/r06aa_kbench_attn_sustained.log 2>&1; cat $out/r06aa_kbench_attn_sustained.log
