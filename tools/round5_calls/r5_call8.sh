# Round 5, call 8: validation of the tree -- smoke, the whole -m gpu suite, the driver's bench command.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05c8; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 1500 python -m pytest tests -x -q -m gpu --durations=8 -p no:cacheprovider > $O/gpu_suite_full.log 2>&1
grep -v "^E    +" $O/gpu_suite_full.log | tail -16
(timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err); python -c "
import json;d=json.load(open('$O/bench_default.json'));print('default',d['value'],d['ms_per_step'],d['median_step']['ms'],d['roofline']['frac'],d['roofline']['avg_launch_ms'],d['config']['select_sweep_launches'],d['parity_vs_oracle']['top100_identical'],d['parity_vs_oracle']['ray_mlp_keys']['max_row_rel_err'],d['scene_setup_s']['ray_mlp_keys_tflops'],d['cpu_baseline']['value'],d['reference_mode']['value'])"
