# Round-5 validation on the GPU box: smoke, the whole -m gpu suite (no -x: every test runs), the driver's bench command, the presets and the stand-in sweep.
#   bash tools/validate_round5.sh <out-name>        (what tools/round5_calls/r5_call13.sh + r5_call15.sh ran; ~17 GPU-minutes)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r05v}; mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 1800 python -m pytest tests -q -m gpu --durations=8 -p no:cacheprovider > $O/gpu_suite_full.log 2>&1
grep -v "^E    +" $O/gpu_suite_full.log | tail -14
(timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err); python -c "
import json;d=json.load(open('$O/bench_default.json'));print('default',d['value'],d['ms_per_step'],d['median_step']['ms'],d['roofline']['frac'],d['roofline']['avg_launch_ms'],d['config']['select_sweep_launches'],d['config']['pipeline'][:24],d['parity_vs_oracle']['top100_identical'],d['scene_setup_s']['ray_mlp_keys_tflops'],d['cpu_baseline']['value'])"
for c in cfg2 cfg3 cfg4; do (timeout 600 python bench.py --config $c --skip-cpu-baseline --skip-reference-mode --l32-steps 0 > $O/bench_$c.json 2> $O/bench_$c.err); python -c "
import json;d=json.load(open('$O/bench_$c.json'));print('$c',d['value'],d['ms_per_step'],d['median_step']['ms'],d['roofline']['frac'],d['roofline']['avg_launch_ms'])"; done
(timeout 900 python bench.py --config cfg5-standin > $O/bench_cfg5_standin.json 2> $O/bench_cfg5_standin.err); python -c "
import json;d=json.load(open('$O/bench_cfg5_standin.json'));print('cfg5',d['value'],d['value_including_scene_setup'],d['value_including_product_scene_setup'],d['parity_summary'])"
