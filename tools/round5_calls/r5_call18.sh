# Round 5, call 18: the headline at 20 steps and the cfg-3 / cfg-4 presets on the final tree.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05c18; mkdir -p $O
(timeout 300 python bench.py --steps 20 --warmup 5 --skip-cpu-baseline --skip-reference-mode --l32-steps 0 > $O/bench_headline.json 2> $O/bench_headline.err)
(timeout 300 python bench.py --steps 20 --warmup 5 --no-pipeline --skip-cpu-baseline --skip-reference-mode --l32-steps 0 > $O/bench_headline_nopipe.json 2> $O/bench_headline_nopipe.err)
for c in cfg3 cfg4; do (timeout 600 python bench.py --config $c --skip-cpu-baseline --skip-reference-mode --l32-steps 0 > $O/bench_$c.json 2> $O/bench_$c.err); done
python - <<PY
import json
for n in ('headline','headline_nopipe','cfg3','cfg4'):
    d=json.load(open('$O/bench_%s.json'%n)); print(n, d['value'], d['ms_per_step'], d['median_step']['ms'], d['median_step']['min_ms'], d['median_step']['max_ms'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['scene_setup_s'].get('ray_mlp_keys_tflops'))
PY
