#!/bin/bash
# round 6, GPU call 2: the software-pipelined attention kernel (attn_pp16s) against attn_pp16mq: bit-identity and time, same box, interleaved
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; export TMPDIR=/tmp; out=gpurun_out; mkdir -p $out
for c in "vitl b32 N3601" "vitl b16 N3601" "N1370" "vitl b1 N3601" "vitb b8 N3601"; do
  KB_S=1 timeout 300 ./tools/kbench attn "$c" ${KB_ITERS:-20}
done > $out/r06b_kbench_attn_s.log 2>&1
cat $out/r06b_kbench_attn_s.log
