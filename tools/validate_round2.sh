set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2k; mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
(timeout 1500 python -m pytest tests -q -m gpu --durations=8 2>&1 | grep -v "^E    +" | tail -60) > $O/gpu_suite.log 2>&1
tail -14 $O/gpu_suite.log
(timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_headline.json 2> $O/bench_headline.err); python -c "
import json;d=json.load(open('$O/bench_headline.json'));print(d['value'],d['ms_per_step'],d['median_step'],d['scene_setup_s'],d['roofline']['achieved'],d['roofline']['frac'],d.get('two_pass_mode',{}).get('value'),d.get('fp32_logits_mode',{}).get('value'))"
for c in cfg2 cfg3 cfg4; do (timeout 400 python bench.py --config $c --skip-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err); python -c "
import json;d=json.load(open('$O/bench_$c.json'));print('$c',d['value'],d['ms_per_step'],d['config']['scoring_path'],d['scene_setup_s'])"; done
