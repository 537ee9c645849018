# Round 5, call 6: the stand-in sweep with ONE arena (no hipMalloc / empty_cache between scenes) and balanced batches; kernel categories of one streamed step
# (stump, 317 M rays, 16 views per step); rocprofv3 kernel trace + PMC passes of the default bench command and of the chain.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05c6; mkdir -p $O
(timeout 900 python bench.py --config cfg5-standin > $O/bench_cfg5_standin.json 2> $O/bench_cfg5_standin.err); tail -3 $O/bench_cfg5_standin.err; python - <<PY
import json
d=json.load(open('$O/bench_cfg5_standin.json'))
print('cfg5', d['value'], d['value_including_scene_setup'], d['value_including_product_scene_setup'], d['scene_setup_s_total'], d['eval_s_total'], d['process_setup_s'], d['parity_summary'])
print(d['scene_setup_breakdown_s_total'])
for r in d['scenes']: print(r['scene'], r['rays'], r['scoring'], r['images_per_step'], r['poses_per_s'], r['sweep_tflops'], r['tokens_per_image_mean'], r['setup_s'], r['step_s'])
PY
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/trace_stump -o trace -- python $GRAFT_REPO_ROOT/bench.py --config cfg5-standin --scenes stump --views-cap 32 --skip-cpu-baseline > $GRAFT_REPO_ROOT/$O/trace_stump.json 2> $GRAFT_REPO_ROOT/$O/trace_stump.err
cd $GRAFT_REPO_ROOT
DB=$(find $O/trace_stump -name "*.db" | head -1); python tools/step_categories.py $DB $O/streamed_step_categories.md; cat $O/streamed_step_categories.md; rm -rf $O/trace_stump
bash tools/profile_round5.sh r05c6/prof > $O/profile.log 2>&1; tail -30 $O/profile.log
