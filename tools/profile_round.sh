#!/bin/bash
# Run on the GPU box (gpurun): rocprofv3 evidence for the bench command of this round, written under gpurun_out/prof_<tag>/.
#   tools/profile_round.sh <tag>
# (every rocprofv3 pass runs under `timeout`: after a GPU fault rocprofv3 can sit in its signal handler until the box's limit - round 5 lost 15 GPU-minutes to that)
# pass 1: --kernel-trace --stats (per-kernel durations);  pass 2/3: --pmc FETCH_SIZE / WRITE_SIZE (separate passes, no trace domains)
tag=${1:-r01}
root=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
out=$root/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
cd $root
CMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-pcie --no-autocast-pass --no-power --no-latency"      # (--no-latency, round 6: batch-32 launches only - a persistent kernel has the same grid at batch 1)
export MOGE_BATCH_SPLIT=0     # one stream: a kernel's trace interval then contains only that kernel
timeout ${PROF_TIMEOUT:-240} rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o tr -- $CMD > $out/trace.log 2>&1
python3 tools/trace_summary.py $out/trace/tr_kernel_trace.csv 60 > $out/kernels_by_grid.csv
cp $out/trace/tr_kernel_stats.csv $out/kernel_stats.csv 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  timeout ${PROF_TIMEOUT:-240} rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$c -o pmc -- $CMD > $out/pmc_$c.log 2>&1
  python3 tools/pmc_summary.py $out/pmc_$c > $out/pmc_$c.csv
done
timeout ${PROF_TIMEOUT:-240} rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT \
  --kernel-trace --output-format csv -d $out/pmc_MFMA -o pmc -- $CMD > $out/pmc_MFMA.log 2>&1
python3 tools/pmc_summary.py $out/pmc_MFMA > $out/pmc_MFMA.csv
sha256sum moge_amd/csrc/gemm_pp.hip > $out/source_hash.txt
grep -h '"metric"' $out/trace.log | tail -1 > $out/bench_under_rocprof.json
rm -rf $out/trace $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $out/pmc_MFMA     # keep the summaries only (raw CSVs are tens of MB)
ls -la $out
