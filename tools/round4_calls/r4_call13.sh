# Round 4, closing call: smoke + the whole -m gpu suite + the default bench line on the final tree.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04v3; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 700 python -m pytest tests -x -q -m gpu --durations=6 -p no:cacheprovider > $O/gpu_suite_full.log 2>&1
grep -v "^E    +" $O/gpu_suite_full.log | tail -14
(timeout 200 python bench.py --skip-reference-mode > $O/bench_default.json 2> $O/bench_default.err); python -c "
import json;d=json.load(open('$O/bench_default.json'));print('default',d['value'],d['ms_per_step'],d['roofline']['frac'],d['config']['images_per_select_sweep_launch'],d['parity_vs_oracle']['top100_identical'],d['parity_vs_oracle']['ray_mlp_keys']['max_row_rel_err'])"
