#!/bin/bash
# Where does the persistent GEMM's energy go?  (--experiments build; run from the root of a tree that holds that build: tools/_ab/exp on the GPU box.)
# For one shape, repeats the kernel with pieces switched off at run time (MOGE_PP_ABL bits: 1 no main-loop DMA, 2 fragment reads only in K-tile 0,
# 4 no epilogue, 8 no MFMA) and samples rocm-smi socket power + shader clock: the energy of a launch is W x ms; differences between rows attribute it.
#   tools/energy_abl.sh <tag> [shapes...]      -> gpurun_out/<tag>_energy_abl.log (relative to $ENERGY_OUT_ROOT or the current directory)
tag=${1:-r05}; shift
shapes=${@:-fc1 fc2.h16}
out=${ENERGY_OUT_ROOT:-.}/gpurun_out/${tag}_energy_abl.log
mkdir -p $(dirname $out)
sample() { while true; do rocm-smi --showpower --showclocks --json 2>/dev/null | python3 -c "
import sys, json, re
try:
    d = json.load(sys.stdin); c = d[sorted(d)[0]]
    p = [float(v) for k, v in c.items() if re.search(r'power', k, re.I) and re.match(r'^[0-9.]+$', str(v))]
    s = [re.search(r'(\d+)Mhz', str(v)) for k, v in c.items() if re.search(r'sclk', k, re.I)]
    s = [int(m.group(1)) for m in s if m]
    print(p[0] if p else -1, s[0] if s else -1)
except Exception: print(-1, -1)
" >> $1; sleep 0.25; done; }
{
  echo "# tools/energy_abl.sh: gemm_pp128p_kernel with pieces switched off (MOGE_PP_ABL), ${EN_ITERS:-2500} launches each; ms = mean launch time, W = rocm-smi socket power"
  echo "# (0.25 s samples, first 2 / last 1 dropped), mJ = W x ms per launch.  ABL bits: 1 no main-loop DMA, 2 fragment reads in K-tile 0 only, 4 no epilogue, 8 no MFMA"
  for s in $shapes; do
    for abl in 0 1 2 4 8 3 7 11 15; do
      : > /tmp/smi.$$; sample /tmp/smi.$$ & smi=$!; sleep 0.6
      MOGE_PP_ABL=$abl KB_EXACT=1 KB_P=1 KB_ROUNDS=1 timeout 120 ./tools/kbench gemm $s ${EN_ITERS:-2500} > /tmp/kb.$$ 2>&1
      kill $smi 2>/dev/null; wait $smi 2>/dev/null
      python3 - "$s" "$abl" /tmp/smi.$$ /tmp/kb.$$ <<'PY'
import sys, re
s, abl, smi, kb = sys.argv[1:5]
rows = [tuple(float(x) for x in l.split()) for l in open(smi) if len(l.split()) == 2]
rows = [r for r in rows if r[0] > 0]; rows = rows[2:-1] if len(rows) > 5 else rows
W = sum(r[0] for r in rows) / max(1, len(rows)); ghz = sum(r[1] for r in rows) / max(1, len(rows)) / 1e3
ms = [float(m.group(1)) for m in re.finditer(r'mean ([0-9.]+) ms', open(kb).read())]
ms = ms[-1] if ms else float('nan')
names = {0: 'full kernel', 1: 'no DMA', 2: 'no fragment reads', 4: 'no epilogue', 8: 'no MFMA', 3: 'no DMA, no reads', 7: 'MFMA + barriers only', 11: 'epilogue + barriers only', 15: 'barriers only'}
print('%-10s abl %2s %-26s %7.3f ms  %7.1f W  %5.2f GHz  %8.1f mJ per launch   (%d samples)' % (s, abl, names.get(int(abl), ''), ms, W, ghz, W * ms, len(rows)))
PY
    done
  done
} > $out 2>&1
cat $out; rm -f /tmp/smi.$$ /tmp/kb.$$
