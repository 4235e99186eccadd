#!/bin/bash
# per-kernel trace of batch-32 steps only, fused ConvTranspose2d + 3x3 off / on (single stream)
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; export TMPDIR=/tmp; out=gpurun_out; mkdir -p $out
CMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-pcie --no-autocast-pass --no-power --no-latency"
export MOGE_BATCH_SPLIT=0
for v in 0 1; do
  rm -rf /tmp/tr$v
  MOGE_FUSE_CT3=$v timeout 240 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr$v -o tr -- $CMD > /tmp/tr$v.log 2>&1
  f=$(find /tmp/tr$v -name "*kernel_trace.csv" | head -1)
  python3 tools/trace_summary.py $f 60 > $out/r06h_kernels_ct3_$v.csv
done
grep "conv_pp\|gemm_pp128p_kernel<4\|gemm_pp128m16_kernel<4\|ct3\|total" $out/r06h_kernels_ct3_0.csv | grep -v ",900,\|,225,\|,128,\|,114," | head -40
echo ======
grep "conv_pp\|gemm_pp128p_kernel<4\|gemm_pp128m16_kernel<4\|ct3\|total" $out/r06h_kernels_ct3_1.csv | grep -v ",900,\|,225,\|,128,\|,114," | head -40
