# Round 5, call 13: validation of the final tree -- smoke, the whole -m gpu suite, the driver's bench command, the stand-in sweep, cfg-2.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05c13; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 1500 python -m pytest tests -x -q -m gpu --durations=8 -p no:cacheprovider > $O/gpu_suite_full.log 2>&1
grep -v "^E    +" $O/gpu_suite_full.log | tail -14
(timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err); python -c "
import json;d=json.load(open('$O/bench_default.json'));print('default',d['value'],d['ms_per_step'],d['median_step']['ms'],d['roofline']['frac'],d['roofline']['avg_launch_ms'],d['config']['select_sweep_launches'],d['parity_vs_oracle']['top100_identical'],d['parity_vs_oracle']['ray_mlp_keys']['max_row_rel_err'],d['scene_setup_s']['ray_mlp_keys_tflops'],d['cpu_baseline']['value'],d['reference_mode']['value'])"
(timeout 900 python bench.py --config cfg5-standin > $O/bench_cfg5_standin.json 2> $O/bench_cfg5_standin.err); tail -2 $O/bench_cfg5_standin.err; python - <<PY
import json
d=json.load(open('$O/bench_cfg5_standin.json'))
print('cfg5', d['value'], d['value_including_scene_setup'], d['value_including_product_scene_setup'], d['scene_setup_s_total'], d['eval_s_total'], d['parity_summary'])
print(d['scene_setup_breakdown_s_total'])
for r in d['scenes']: print(r['scene'], r['scoring'], r['images_per_step'], r['poses_per_s'], r['sweep_tflops'], r['tokens_per_image_mean'], r['setup_s'], r['step_s'])
PY
(timeout 300 python bench.py --config cfg2 --steps 20 --skip-cpu-baseline --skip-reference-mode --l32-steps 0 > $O/bench_cfg2.json 2> $O/bench_cfg2.err); python -c "
import json;d=json.load(open('$O/bench_cfg2.json'));print('cfg2',d['value'],d['ms_per_step'],d['median_step'],d['roofline']['avg_launch_ms'])"
