# Round 4, third GPU call: the tests the second call lost to a hang (fixed), the new kernels' parity, the repaired ablation power table, cfg2 / T&T timings.
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r4c3; mkdir -p $O
timeout 900 python -m pytest -q -m gpu -p no:cacheprovider -x --durations=6 tests/test_gpu_parity.py -k "topk" tests/test_gpu_select.py tests/test_gpu_integration_stub.py tests/test_isocell_module.py tests/test_gpu_rccl_single.py 2>&1 | grep -v "^E    +" | tail -25 > $O/tests_a.log; tail -12 $O/tests_a.log
timeout 700 python -m pytest -q -m gpu -p no:cacheprovider -x --durations=6 tests/test_gpu_e2e.py -k "two_ranks or stand_in" "tests/test_gpu_configs.py::test_headline_500k_x64_scores_and_top100_against_the_oracle" -s 2>&1 | grep -E "passed|failed|error|Error|key parity|headline|assert" | tail -20 > $O/tests_b.log; tail -12 $O/tests_b.log
for a in abl2 abl18 abl27 abl59; do
  SIXDGS_LIB=$PWD/build/variants/lib_abl.so POWER_RAYS=16000000 timeout 150 python tools/power_trace.py $a 6 2>&1 | grep -E "^POWER" >> $O/power_ablation.log
done
cut -c1-330 $O/power_ablation.log
(timeout 300 python bench.py --config cfg2 --steps 20 --warmup 3 --skip-cpu-baseline --skip-reference-mode --l32-steps 0 > $O/bench_cfg2.json 2> $O/bench_cfg2.err); python -c "import json;d=json.load(open('$O/bench_cfg2.json'));print('cfg2', d['value'], d['ms_per_step'], d['median_step'], d['roofline']['avg_launch_ms'])"
(timeout 400 python bench.py --config cfg5-standin --scenes tt_ --skip-cpu-baseline > $O/bench_cfg5_tt.json 2> $O/bench_cfg5_tt.err); python - <<PY
import json
d=json.load(open("$O/bench_cfg5_tt.json"))
print("cfg5 tt", d["value"], d["roofline"]["frac"])
for r in d["scenes"]: print(r["scene"], r["tokens_per_image_mean"], r["eval_s"], r["poses_per_s"], r["sweep_tflops"])
PY
