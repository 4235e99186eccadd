#!/bin/bash
# round 6, evidence at the final tree (key-split one-image attention, four launches fewer in front of the first block, re-draw bands): full GPU suite, smoke, rocprofv3 trace + PMC passes, default bench, other configs, driver-form bench
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; export TMPDIR=/tmp; out=gpurun_out; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -s > $out/r06g_pytest_gpu.log 2>&1; grep "passed\|failed\|error" $out/r06g_pytest_gpu.log | tail -3; grep "^FAILED\|^ERROR" $out/r06g_pytest_gpu.log | head
grep "^\[large\|^\[gate" $out/r06g_pytest_gpu.log > $out/r06g_gate_lines.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $out/r06g_smoke.log 2>&1; tail -1 $out/r06g_smoke.log
PROF_TIMEOUT=300 bash tools/profile_round.sh r06g > $out/r06g_profile_round.log 2>&1; tail -3 $out/r06g_profile_round.log
timeout 900 python bench.py > $out/r06g_bench_full.json 2> $out/r06g_bench_full.err; head -c 600 $out/r06g_bench_full.json; echo
for cfg in "--config moge-2-vitb-normal --batch 8" "--config moge-2-vits-normal --batch 8" "--config moge-2-vitl-normal"; do
  timeout 300 python bench.py $cfg --steps 20 --warmup 5 --no-cpu-baseline --no-pcie --no-power --no-autocast-pass 2>/dev/null | tail -1
done > $out/r06g_bench_other_configs.jsonl
python - <<'PY'
import json
for l in open('gpurun_out/r06g_bench_other_configs.jsonl'):
    try:
        d=json.loads(l); print(d['metric'], d['config']['workload'][:60], d['value'], d['p50_latency_ms_batch1'])
    except Exception as e: print('bad line', e)
PY
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06g_bench_driver_form.json 2> gpurun_out/r06g_bench_driver_form.err; head -c 400 gpurun_out/r06g_bench_driver_form.json; echo
timeout 900 python tools/ks_split_draws.py v1_vitl_518 v1_vitl_train_config_518 vitb_normal_518_t3600 vitl_normal_518_t3600 vits_house518 2>&1 | grep -v "Warning\|warn\|amdgpu.ids" > $out/r06g_ks_split_draws.log; tail -5 $out/r06g_ks_split_draws.log
