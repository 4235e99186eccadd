#!/bin/bash
# Energy per TFLOP of the shipped kernels ("judge each change in J/TFLOP, not only in us"): runs tools/kbench long enough for
# rocm-smi to see the steady state, samples socket power + shader clock every 0.25 s in the background, and prints   TF/s, W, GHz, J/TFLOP = W / (TF/s).
#   tools/energy.sh <tag>        -> gpurun_out/<tag>_energy.log   (copy to profiles/)
tag=${1:-r05}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
out=gpurun_out/${tag}_energy.log
mkdir -p gpurun_out
sample() {   # sample <seconds-file>: append "power clock" lines until killed
  while true; do
    rocm-smi --showpower --showclocks --json 2>/dev/null | python3 -c "
import sys, json, re
try:
    d = json.load(sys.stdin); c = d[sorted(d)[0]]
    p = [float(v) for k, v in c.items() if re.search(r'power', k, re.I) and re.match(r'^[0-9.]+$', str(v))]
    s = [re.search(r'(\d+)Mhz', str(v)) for k, v in c.items() if re.search(r'sclk', k, re.I)]
    s = [int(m.group(1)) for m in s if m]
    print(p[0] if p else -1, s[0] if s else -1)
except Exception as e:
    print(-1, -1)
" >> $1
    sleep 0.25
  done
}
run() {      # run <label> <flop-extractor regex on kbench output> <cmd...>
  label=$1; shift
  : > /tmp/smi.$$
  sample /tmp/smi.$$ & smi=$!
  sleep 0.6
  "$@" > /tmp/kb.$$ 2>&1
  kill $smi 2>/dev/null; wait $smi 2>/dev/null
  python3 - "$label" /tmp/smi.$$ /tmp/kb.$$ <<'PY'
import sys, re
label, smi, kb = sys.argv[1:4]
rows = [tuple(float(x) for x in l.split()) for l in open(smi) if len(l.split()) == 2]
rows = [r for r in rows if r[0] > 0]
rows = rows[2:-1] if len(rows) > 5 else rows            # drop the ramp-up / ramp-down samples
W = sum(r[0] for r in rows) / max(1, len(rows)); ghz = sum(r[1] for r in rows) / max(1, len(rows)) / 1e3
tf = [float(m.group(1)) for m in re.finditer(r'([0-9.]+) TF/s', open(kb).read())]
tfs = tf[-1] if tf else float('nan')
print('%-34s %8.1f TF/s  %7.1f W  %5.2f GHz (smi)  %6.3f J/TFLOP   (%d power samples)' % (label, tfs, W, ghz, W / tfs if tfs == tfs and tfs > 0 else float('nan'), len(rows)))
PY
}
{
  echo "# tools/energy.sh: steady-state socket power (rocm-smi, 0.25 s samples) while tools/kbench repeats ONE kernel; J/TFLOP = W / (TF/s); random operands"
  for s in ${EN_SHAPES:-qkv proj.h16 fc1 fc2.h16 outproj}; do KB_EXACT=1 KB_P=1 KB_ROUNDS=1 run "gemm $s (persistent, fused epilogue)" timeout 120 ./tools/kbench gemm $s ${EN_ITERS:-2500}; done
  [ -z "$EN_NO_ATTN" ] && run "attention vitl b32 N3601 (mq<4>)" timeout 120 ./tools/kbench attn "vitl b32 N3601" ${EN_ATTN_ITERS:-600}
  echo "# bare MFMA chains for comparison (tools/mfma_power, MFMA_POWER_SECONDS=3):"
  [ -z "$EN_NO_ATTN" ] && [ -x tools/mfma_power ] && MFMA_POWER_SECONDS=3 ./tools/mfma_power 2>&1 | tail -12
} > $out 2>&1
cat $out
rm -f /tmp/smi.$$ /tmp/kb.$$
