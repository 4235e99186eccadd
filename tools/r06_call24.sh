#!/bin/bash
# --experiments build: the persistent GEMM's epilogue split at the next-tile prefetch (before / prefetch / behind), qkv fc1 proj.h16 fc2.h16
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; export TMPDIR=/tmp; out=gpurun_out; mkdir -p $out
for f in qkv fc1 proj.h16 fc2.h16; do KB_EXACT=1 KB_TS=1 KB_P=1 timeout 200 ./tools/kbench gemm $f 10 2>&1 | grep "ts wave\|pp128p" ; done > $out/r06u_pp128p_epilogue_split.log 2>&1; cat $out/r06u_pp128p_epilogue_split.log
