# Round 5, call 5: the presets on the round-5 build (fold, packing, pipeline): cfg5 stand-in at scale with itemised set-up, cfg2, cfg3, cfg4; the two new e2e tests.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05c5; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_bench_contract.py -x -q -m gpu -p no:cacheprovider -k "streamed_inference or eight_ranks_on_one_gpu or a24" > $O/new_tests.log 2>&1; grep -v "^E    +" $O/new_tests.log | tail -12
(timeout 900 python bench.py --config cfg5-standin > $O/bench_cfg5_standin.json 2> $O/bench_cfg5_standin.err); python - <<PY
import json
d=json.load(open('$O/bench_cfg5_standin.json'))
print('cfg5', d['value'], d['value_including_scene_setup'], d['value_including_product_scene_setup'], d['scene_setup_s_total'], d['eval_s_total'], d['parity_summary'])
print(d['scene_setup_breakdown_s_total'])
for r in d['scenes']: print(r['scene'], r['rays'], r['scoring'], r['poses_per_s'], r['sweep_tflops'], r['tokens_per_image_mean'], r['setup_s'], r['setup_breakdown_s'])
PY
for C in cfg2 cfg3 cfg4; do
  (timeout 600 python bench.py --config $C --skip-cpu-baseline --skip-reference-mode --l32-steps 0 > $O/bench_$C.json 2> $O/bench_$C.err)
  python -c "
import json;d=json.load(open('$O/bench_$C.json'));print('$C', d['value'],d['ms_per_step'],d['median_step']['ms'],d['roofline']['frac'],d['roofline']['avg_launch_ms'],d['scene_setup_s'].get('ray_mlp_keys_tflops'), d['config'].get('pipeline','')[:20])"
done
