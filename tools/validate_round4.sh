# Round-4 validation on the GPU box: smoke, the whole -m gpu suite, the default bench line, the presets.  bash tools/validate_round4.sh <out-name>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r04v}; mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 2400 python -m pytest tests -q -m gpu --durations=10 -p no:cacheprovider > $O/gpu_suite_full.log 2>&1
grep -v "^E    +" $O/gpu_suite_full.log | tail -40 > $O/gpu_suite.log; tail -16 $O/gpu_suite.log
(timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err); python - <<PY
import json
d=json.load(open("$O/bench_default.json"))
print("default", d["value"], d["ms_per_step"], d["median_step"]["ms"], d["scene_setup_s"], d["roofline"]["achieved"], d["roofline"]["frac"], d.get("two_pass_mode",{}).get("value"), d.get("fp32_logits_mode",{}).get("value"))
print("parity", {k: d["parity_vs_oracle"].get(k) for k in ("top100_identical","score_rel_err","rot_err_deg","trans_err")}, d["cpu_baseline"]["value"], d["cpu_baseline"].get("reference_cost_per_pose"), d["reference_mode"]["value"])
PY
(timeout 300 python bench.py --steps 20 --warmup 5 --skip-cpu-baseline --skip-reference-mode > $O/bench_headline.json 2> $O/bench_headline.err); python -c "
import json;d=json.load(open('$O/bench_headline.json'));print('headline20',d['value'],d['ms_per_step'],d['median_step'],d['roofline']['achieved'],d['roofline']['frac'],d['roofline']['avg_launch_ms'])"
for c in cfg2 cfg3 cfg4; do (timeout 500 python bench.py --config $c --skip-cpu-baseline --skip-reference-mode > $O/bench_$c.json 2> $O/bench_$c.err); python -c "
import json;d=json.load(open('$O/bench_$c.json'));print('$c',d['value'],d['ms_per_step'],d['config']['scoring_path'],d['scene_setup_s']['total'],d['scene_setup_s']['key_plane_buffer_alloc'],d['scene_setup_s']['ray_mlp_keys'],d['roofline']['frac'])"; done
