#!/bin/bash
# Run on the GPU box: tools/mfma_power (bare MFMA accumulate chains, operands in registers, no data movement) with rocm-smi power / clock
# samples taken WHILE each configuration runs.  Output: gpurun_out/<tag>_mfma_power.log  (copy to profiles/).
#   tools/mfma_power.sh <tag>
tag=${1:-r04}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
out=gpurun_out/${tag}_mfma_power.log
mkdir -p gpurun_out
[ -x tools/mfma_power ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/mfma_power.cpp -o tools/mfma_power
{
  echo "# tools/mfma_power.cpp: 1024 workgroups x {4, 8} waves, each wave a chain of 4000 x {32 v_mfma_f32_32x32x16_f16 | 64 v_mfma_f32_16x16x32_f16}"
  echo "# MFMA_POWER_SECONDS=4: every configuration is repeated for ~4 s; rocm-smi is sampled every 0.5 s in the background (power = average socket power)"
  rocm-smi --showproductname 2>/dev/null | grep -i "card series\|GFX" | head -3
  ( while true; do
      j=$(rocm-smi --showpower --showclocks --json 2>/dev/null | tr -d '\n')
      echo "    [smi t=$(date +%s.%N | cut -c1-14)] $j" | python3 -c "
import sys, json, re
ln = sys.stdin.read()
m = re.search(r'(\{.*\})', ln)
head = ln[:ln.index('{')] if m else ln
try:
    d = json.loads(m.group(1)); c = d[sorted(d)[0]]
    keep = {k: v for k, v in c.items() if re.search(r'power|sclk', k, re.I)}
    print(head + json.dumps(keep))
except Exception:
    print(ln.strip()[:300])
"
      sleep 0.5
    done ) &
  smi=$!
  MFMA_POWER_SECONDS=4 ./tools/mfma_power
  kill $smi 2>/dev/null
} > $out 2>&1
cat $out
