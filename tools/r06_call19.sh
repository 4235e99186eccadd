#!/bin/bash
# --experiments build: phase timeline of the two-workgroups-per-CU GEMM (PP_EXP 6) and its one-workgroup sibling (PP_EXP 5) next to the persistent product kernel's per-tile stamps; stream-K attention tests
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; export TMPDIR=/tmp; out=gpurun_out; mkdir -p $out
for f in qkv fc1; do
  KB_EXACT=1 KB_TS=1 KB_TS_DIR=$out KB_EXP=1 KB_ROUNDS=2 timeout 300 ./tools/kbench gemm $f 10 > $out/r06q_kbench_gemm_occ2_timeline_$f.log 2>&1
  grep -v "^   ts wave" $out/r06q_kbench_gemm_occ2_timeline_$f.log | tail -14
  for e in 5 6; do echo "== $f PP_EXP $e"; python3 tools/pp64_timeline.py $out/pp64_timeline_${f}_exp$e.csv 1024; done
done > $out/r06q_occ2_timeline.log 2>&1; cat $out/r06q_occ2_timeline.log
for f in qkv fc1; do KB_EXACT=1 KB_TS=1 KB_P=1 timeout 200 ./tools/kbench gemm $f 10 2>&1 | grep "ts wave\|pp128p" ; done > $out/r06q_pp128p_tile_timeline.log 2>&1; cat $out/r06q_pp128p_tile_timeline.log
gzip -f $out/pp64_timeline_*.csv
true
