# Round-6 validation on the GPU box: smoke, the whole -m gpu suite (no -x: every test runs), the driver's bench command, the presets and the stand-in sweep.
#   bash tools/validate_round6.sh <out-name>        (~20 GPU-minutes)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r06v}; mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
( time timeout 1800 python -m pytest tests -q -m gpu --durations=12 -p no:cacheprovider 2>&1 | grep -v "^E    +" | tail -n 40 ) > $O/gpu_suite.log 2>&1
tail -n 24 $O/gpu_suite.log
(timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err); python -c "
import json;d=json.load(open('$O/bench_default.json'));print('default',d['value'],d['ms_per_step'],d['median_step']['ms'],d['roofline']['frac'],d['roofline']['avg_launch_ms'],d['config']['select_sweep_launches'],d['config']['pipeline'][:24],d['parity_vs_oracle']['top100_identical'],d['scene_setup_s']['ray_mlp_keys_tflops'],d['cpu_baseline']['value'],d['headline_b4']['value'],d['reference_mode']['value'])"
for c in cfg2 cfg3 cfg4; do (timeout 600 python bench.py --config $c --skip-cpu-baseline --skip-reference-mode --l32-steps 0 > $O/bench_$c.json 2> $O/bench_$c.err); python -c "
import json;d=json.load(open('$O/bench_$c.json'));print('$c',d['value'],d['ms_per_step'],d['median_step']['ms'],d['roofline']['frac'],d['roofline']['avg_launch_ms'],d['scene_setup_s']['total'])"; done
(timeout 400 python bench.py --mode reference --batch 16 --steps 20 --skip-cpu-baseline > $O/bench_reference_mode_16.json 2> $O/bench_reference_mode_16.err); python -c "
import json;d=json.load(open('$O/bench_reference_mode_16.json'));print('reference mode, 16 images per step',d['value'],d['ms_per_step'])"
(timeout 900 python bench.py --config cfg5-standin > $O/bench_cfg5_standin.json 2> $O/bench_cfg5_standin.err); python -c "
import json;d=json.load(open('$O/bench_cfg5_standin.json'));print('cfg5',d['value'],d['value_including_scene_setup'],d['value_including_product_scene_setup'],d['parity_summary'])"
python -W ignore tools/time_image_side.py 2>&1 | grep -v amdgpu.ids > $O/image_side.md; cat $O/image_side.md
