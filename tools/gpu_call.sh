#!/bin/bash
# One gpurun call: tools/gpu_call.sh <tag> <what...>   (what: tests kb_rb kb_conv kb_gemm kb_attn bench ab_conv ...)
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
export TMPDIR=/tmp
out=gpurun_out
mkdir -p $out
for what in "$@"; do
  case $what in
    mfma_power) bash tools/mfma_power.sh $tag > /dev/null 2>&1; grep -v "smi t=" $out/${tag}_mfma_power.log | tail -20 ;;
    tests_s)    timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider -s > $out/${tag}_pytest_s.log 2>&1; grep "passed\|failed\|error" $out/${tag}_pytest_s.log | tail -5; grep "^FAILED\|^ERROR\|Error" $out/${tag}_pytest_s.log | head -20 ;;
    kb_resid)   KB_PP=1 KB_ROUNDS=3 timeout 400 ./tools/kbench gemm proj 10 > $out/${tag}_kbench_resid.log 2>&1; KB_PP=1 KB_ROUNDS=3 timeout 400 ./tools/kbench gemm fc2 10 >> $out/${tag}_kbench_resid.log 2>&1; grep -v "^ " $out/${tag}_kbench_resid.log | grep -v "vit\|b1\.\|b4\." | head -60 ;;
    tests_half) timeout 900 python -m pytest tests/test_hip_gemm_pp.py tests/test_hip_parity.py tests/test_hip_v1.py tests/test_hip_kernels.py -m gpu -q -p no:cacheprovider -s -x > $out/${tag}_pytest_half.log 2>&1; grep "passed\|failed\|error" $out/${tag}_pytest_half.log | tail -5; grep "^FAILED\|^ERROR\|Error\|assert" $out/${tag}_pytest_half.log | head -20 ;;
    tests_gemm) timeout 900 python -m pytest tests/test_hip_gemm_pp.py -m gpu -q -p no:cacheprovider > $out/${tag}_pytest_gemm.log 2>&1; tail -3 $out/${tag}_pytest_gemm.log ;;
    kb_ab)      # two library builds on one box, alternating: tools/_ab/old = the previous commit's build (same tree layout), ./ = this tree
                for r in 1 2; do for f in ${KB_AB_SHAPES:-qkv proj.h16 fc1 fc2.h16 outproj proj+fold}; do
                  for w in tools/_ab/old .; do echo "== $w round $r"; (cd $w && KB_P=1 KB_ROUNDS=2 timeout 120 ./tools/kbench gemm $f 10) | grep "interleaved\|bad [1-9]" | grep -v "b1\.\|b4\.\|vit"; done
                done; done > $out/${tag}_kbench_gemm_ab.log 2>&1; cat $out/${tag}_kbench_gemm_ab.log ;;
    kb_ab_attn) # attention A/B over library builds on one box, alternating: tools/_ab/old (the previous build, same tree layout) and ./
                for r in 1 2; do for w in ${KB_AB_DIRS:-tools/_ab/old .}; do echo "== $w round $r"; (cd $w && timeout 120 ./tools/kbench attn "${KB_ATTN_CASE:-vitl b}" ${KB_ITERS:-30}) | grep -v "pp16 "; done; done > $out/${tag}_kbench_attn_ab.log 2>&1; cat $out/${tag}_kbench_attn_ab.log ;;
    kb_ab_lat)  # latency-regime GEMMs, old / new library alternating on one box
                for r in 1 2; do for f in ${KB_AB_SHAPES:-b1.proj b1.fc2 vitb1.}; do for w in ${KB_AB_DIRS:-tools/_ab/old .}; do echo "== $w round $r"; (cd $w && KB_LAT=1 KB_ROUNDS=2 timeout 120 ./tools/kbench gemm $f 20) | grep "interleaved\|bad [1-9]"; done; done; done > $out/${tag}_kbench_gemm_lat_ab.log 2>&1; cat $out/${tag}_kbench_gemm_lat_ab.log ;;
    kb_m3)      for f in qkv proj fc1 fc2 outproj; do KB_M3=1 KB_ROUNDS=3 timeout 300 ./tools/kbench gemm $f 10; done > $out/${tag}_kbench_gemm_m3.log 2>&1; grep -v "^   ts" $out/${tag}_kbench_gemm_m3.log | grep -v "b1\.\|b4\.\|vit" ;;
    tests_model) timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_v1.py tests/test_hip_multiproc.py -m gpu -q -p no:cacheprovider -s > $out/${tag}_pytest_model.log 2>&1; grep "passed\|failed\|error" $out/${tag}_pytest_model.log | tail -5; grep "^FAILED\|^ERROR\|Error\|assert" $out/${tag}_pytest_model.log | head -20 ;;
    tests)      timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $out/${tag}_pytest.log 2>&1; tail -25 $out/${tag}_pytest.log ;;
    tests_new)  timeout 600 python -m pytest tests/test_hip_conv_ex.py tests/test_hip_parity.py -m gpu -q -p no:cacheprovider -s > $out/${tag}_pytest_new.log 2>&1; tail -30 $out/${tag}_pytest_new.log ;;
    kb_rb)      KB_ROUNDS=3 timeout 300 ./tools/kbench rb - 10 > $out/${tag}_kbench_rb.log 2>&1; cat $out/${tag}_kbench_rb.log ;;
    kb_conv)    timeout 300 ./tools/kbench conv - 10 > $out/${tag}_kbench_conv.log 2>&1; cat $out/${tag}_kbench_conv.log ;;
    kb_gemm)    KB_PP=1 KB_ROUNDS=3 timeout 400 ./tools/kbench gemm - 10 > $out/${tag}_kbench_gemm.log 2>&1; grep -v "^ " $out/${tag}_kbench_gemm.log | head -80 ;;
    kb_attn)    timeout 300 ./tools/kbench attn - 10 > $out/${tag}_kbench_attn.log 2>&1; cat $out/${tag}_kbench_attn.log ;;
    bench)      timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pcie > $out/${tag}_bench.json 2> $out/${tag}_bench.err; cat $out/${tag}_bench.json ;;
    bench_full) timeout 900 python bench.py > $out/${tag}_bench_full.json 2> $out/${tag}_bench_full.err; cat $out/${tag}_bench_full.json ;;
    ab_conv)    # same box, alternating: new conv forms off / on
                for r in 1 2; do
                  for v in 0 1; do
                    MOGE_CONV_RB=$v MOGE_CONV_SIDE_REG=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pcie 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); k = d['kernel_classes']
print('conv forms $v: %.1f img/s  %.2f ms/step  conv %.2f ms  gemm_pp %.2f  attn %.2f  post %.2f' % (d['value'], d['ms_per_step'], k['conv']['ms_per_step'], k['gemm_pp']['ms_per_step'], k['attn']['ms_per_step'], k['post']['ms_per_step']))"
                  done
                done > $out/${tag}_ab_conv.log 2>&1; cat $out/${tag}_ab_conv.log ;;
    kb_corun)   timeout 300 ./tools/kbench corun - 10 > $out/${tag}_kbench_corun.log 2>&1; cat $out/${tag}_kbench_corun.log ;;
    kb_rb_ts)   KB_TS=1 KB_ROUNDS=1 timeout 300 ./tools/kbench rb - 5 > $out/${tag}_kbench_rb_ts.log 2>&1; cat $out/${tag}_kbench_rb_ts.log ;;
    corun_trace) rm -rf /tmp/ct; rocprofv3 --kernel-trace --output-format csv -d /tmp/ct -o ct -- ./tools/kbench corun fc2 4 > $out/${tag}_corun_trace.log 2>&1
                f=$(find /tmp/ct -name "*kernel_trace.csv" | head -1); python3 tools/corun_overlap.py $f >> $out/${tag}_corun_trace.log 2>&1; tail -40 $out/${tag}_corun_trace.log ;;
    pcie)       timeout 600 python tools/pcie_check.py > $out/${tag}_pcie_check.log 2>&1; cat $out/${tag}_pcie_check.log ;;
    attn_q4)    timeout 300 ./tools/kbench attn - 10 > $out/${tag}_kbench_attn.log 2>&1; cat $out/${tag}_kbench_attn.log
                MOGE_ATTN_KERN=2 timeout 300 python -m pytest tests/test_hip_kernels.py -k attention -q -p no:cacheprovider 2>&1 | tail -3 ;;
    kb_rb_var)  for v in ${RBVARS:-0 4 8 12}; do echo "== CONV_RB_VAR $v"; KB_RBVAR=$v KB_TS=1 KB_ROUNDS=2 timeout 300 ./tools/kbench rb - 10 2>&1 | grep -v "tile [0-2]:\|b16\|grid b8\|odd" ; done > $out/${tag}_kbench_rb_var.log 2>&1; cat $out/${tag}_kbench_rb_var.log ;;
    ab)         # same box, alternating: MOGE_$AB_VAR = 0 / 1 (e.g. AB_VAR=L4DOT)
                for r in 1 2; do
                  for v in ${AB_VALS:-0 1}; do
                    env MOGE_${AB_VAR}=$v timeout 300 python bench.py ${BENCH_ARGS} --steps 10 --warmup 3 --no-cpu-baseline --no-pcie 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); k = d['kernel_classes']
print('$AB_VAR=$v: %.1f img/s  %.2f ms/step  p50(B=1) %.2f  conv %.2f ms  gemm_pp %.2f  attn %.2f  post %.2f  norm %.2f' % (d['value'], d['ms_per_step'], d['p50_latency_ms_batch1'], k['conv']['ms_per_step'], k['gemm_pp']['ms_per_step'], k['attn']['ms_per_step'], k['post']['ms_per_step'], k['norm']['ms_per_step']))"
                  done
                done > $out/${tag}_ab_${AB_VAR}${AB_SUFFIX}.log 2>&1; cat $out/${tag}_ab_${AB_VAR}${AB_SUFFIX}.log ;;
    tests_conv) timeout 600 python -m pytest tests/test_hip_conv_ex.py tests/test_hip_parity.py -m gpu -q -p no:cacheprovider > $out/${tag}_pytest_conv.log 2>&1; tail -15 $out/${tag}_pytest_conv.log ;;
    other_cfgs) for a in "--config moge-2-vitb-normal --batch 8" "--config moge-2-vitl-normal" "--shape mixed --config moge-2-vitl-normal" "--num-tokens 1369" "--config moge-2-vits-normal --batch 8"; do
                  timeout 400 python bench.py $a --steps 10 --warmup 3 --no-cpu-baseline --no-pcie 2>/dev/null | tail -1
                done > $out/${tag}_bench_other_configs.jsonl; python -c "
import json
for l in open('$out/${tag}_bench_other_configs.jsonl'):
    d = json.loads(l); k = d['kernel_classes']
    print(d['metric'][:70], '| %.1f img/s %.2f ms/step p50 %.2f | conv %.2f gemm_pp %.2f gemm %.2f attn %.2f post %.2f' % (d['value'], d['ms_per_step'], d['p50_latency_ms_batch1'], k['conv']['ms_per_step'], k['gemm_pp']['ms_per_step'], k['gemm']['ms_per_step'], k['attn']['ms_per_step'], k['post']['ms_per_step']))" ;;
    kb_gemm_mrg) for f in qkv proj fc1 fc2 outproj tailM; do KB_PP=1 KB_ROUNDS=3 timeout 300 ./tools/kbench gemm $f 10; done > $out/${tag}_kbench_gemm_mrg.log 2>&1; grep -v "^   ts" $out/${tag}_kbench_gemm_mrg.log | grep -v "b1\.\|b4\.\|vits" ;;
    tests_gemm) MOGE_PP_KERN=${PPK:-3} timeout 600 python -m pytest tests/test_hip_gemm_pp.py -m gpu -q -p no:cacheprovider 2>&1 | tail -5 ;;
    batch_sweep) for r in 1 2; do for b in ${BATCHES:-32 36 18 27}; do
                  timeout 400 python bench.py --batch $b --steps 8 --warmup 3 --no-cpu-baseline --no-pcie 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); k = d['kernel_classes']
print('batch $b: %.1f img/s  %.2f ms/step  (kernel sum single-stream %.2f ms: gemm_pp %.2f attn %.2f conv %.2f)' % (d['value'], d['ms_per_step'], d['whole_path']['kernel_ms_per_step'], k['gemm_pp']['ms_per_step'], k['attn']['ms_per_step'], k['conv']['ms_per_step']))"
                done; done > $out/${tag}_batch_sweep.log 2>&1; cat $out/${tag}_batch_sweep.log ;;
    tests_mp)   timeout 600 python -m pytest tests/test_hip_multiproc.py -m gpu -q -p no:cacheprovider 2>&1 | tail -5 ;;
    ab_q4)      for cfg in "vitb8:--config moge-2-vitb-normal --batch 8" "vitl8:--batch 8" "vitl4:--batch 4" "vitl16:--batch 16"; do
                  echo "== ${cfg%%:*}"; AB_SUFFIX=_${cfg%%:*} BENCH_ARGS="${cfg#*:}" AB_VAR=ATTN_Q2_MAX_WGS AB_VALS="768 2900" bash tools/gpu_call.sh $tag ab; done ;;
    lat)        # batch-1 anatomy: kernel classes from the HIP-event profiler + a rocprofv3 kernel trace of the same command
                timeout 300 python bench.py --batch 1 --steps 30 --warmup 5 --no-cpu-baseline --no-pcie > $out/${tag}_bench_b1.json 2>/dev/null
                python -c "
import json
d = json.loads(open('$out/${tag}_bench_b1.json').read().strip().splitlines()[-1]); k = d['kernel_classes']
print('B=1: %.1f img/s %.3f ms/step p50 %.3f | ' % (d['value'], d['ms_per_step'], d['p50_latency_ms_batch1']) + ' '.join('%s %.3f' % (n, v['ms_per_step']) for n, v in k.items()))"
                rm -rf /tmp/lt; rocprofv3 --kernel-trace --output-format csv -d /tmp/lt -o lt -- python bench.py --batch 1 --steps 30 --warmup 5 --no-cpu-baseline --no-pcie --no-profile > /dev/null 2>&1
                python3 tools/trace_summary.py /tmp/lt/lt_kernel_trace.csv 80 > $out/${tag}_b1_kernels_by_grid.csv; head -60 $out/${tag}_b1_kernels_by_grid.csv ;;
    kb_lat)     for f in b1. b4.; do KB_LAT=1 KB_ROUNDS=3 timeout 300 ./tools/kbench gemm $f 20; done > $out/${tag}_kbench_gemm_latency.log 2>&1; grep -v "^   ts" $out/${tag}_kbench_gemm_latency.log ;;
    ab_heads)   for cfg in "vitl1:--batch 1" "vitl4:--batch 4" "vitb8:--config moge-2-vitb-normal --batch 8" "vitln1:--config moge-2-vitl-normal --batch 1" "vitln4:--config moge-2-vitl-normal --batch 4"; do
                  echo "== ${cfg%%:*}"; AB_SUFFIX=_${cfg%%:*} BENCH_ARGS="${cfg#*:}" AB_VAR=HEAD_STREAMS AB_VALS="0 1" bash tools/gpu_call.sh $tag ab; done
                echo "== vitl32 (HEAD_STREAMS_MAX_B 7 / 16)"; AB_SUFFIX=_vitl32 BENCH_ARGS="" AB_VAR=HEAD_STREAMS_MAX_B AB_VALS="7 16" bash tools/gpu_call.sh $tag ab ;;
    tests_par)  timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -5 ;;
    tests_rec)  timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_hip_parity.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -5 ;;
    tests_cv)   timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_hip_conv_ex.py tests/test_hip_parity.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -5 ;;
    kb_attn_b1) for f in "b1 N" "b2 N"; do timeout 300 ./tools/kbench attn "$f" 20; done > $out/${tag}_kbench_attn_b1.log 2>&1; cat $out/${tag}_kbench_attn_b1.log ;;
    ab1)        # batch-1 A/B of MOGE_$AB_VAR over $AB_VALS
                AB_SUFFIX=_b1 BENCH_ARGS="--batch 1" bash tools/gpu_call.sh $tag ab ;;
    ab2)        AB_SUFFIX=_b2 BENCH_ARGS="--batch 2" bash tools/gpu_call.sh $tag ab ;;
    tests_new2) timeout 600 python -m pytest tests/test_hip_kernels.py tests/test_hip_parity.py -m gpu -q -p no:cacheprovider -k "cast or half_model" 2>&1 | tail -3 ;;
    tests_attn) timeout 600 python -m pytest tests/test_hip_kernels.py -k attention -m gpu -q -p no:cacheprovider -x 2>&1 | tail -8 ;;
    kb_attn_x)  # attn_pp16x_kernel (ping-pong) against attn_pp16mq: interleaved in one process per shape, bit-identity checked by kbench
                for f in ${KB_ATTN_CASES:-"vitl b32 N3601" "vitl b16 N3601" "518x1036"}; do KB_X=1 timeout 120 ./tools/kbench attn "$f" ${KB_ITERS:-20}; done > $out/${tag}_kbench_attn_x.log 2>&1; cat $out/${tag}_kbench_attn_x.log ;;
    *) echo "unknown step $what" ;;
  esac
done
