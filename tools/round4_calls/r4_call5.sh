# Round 4, fifth GPU call: the round's validation (smoke, whole GPU suite, bench presets), the profile passes, graph-mode cfg2, the cfg5 stand-in at full scale.
cd $GRAFT_REPO_ROOT
bash tools/validate_round4.sh r04v
bash tools/profile_round4.sh r04p > gpurun_out/r04p_stdout.log 2>&1; tail -30 gpurun_out/r04p_stdout.log
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04v
(timeout 200 python bench.py --config cfg2 --graph --steps 20 --warmup 3 --skip-cpu-baseline --skip-reference-mode --l32-steps 0 > $O/bench_cfg2_graph.json 2> $O/bench_cfg2_graph.err); python -c "import json;d=json.load(open('$O/bench_cfg2_graph.json'));print('cfg2 graph', d['value'], d['ms_per_step'], d['median_step'])"
(timeout 900 python bench.py --config cfg5-standin > $O/bench_cfg5_standin.json 2> $O/bench_cfg5_standin.err); python - <<PY
import json
d=json.load(open("$O/bench_cfg5_standin.json"))
print("cfg5", d["value"], d["value_including_scene_setup"], d["parity_summary"], d["roofline"]["frac"], d["roofline"]["ray_mlp_chain_tflops"])
for r in d["scenes"]: print(r["scene"], r["rays"], r["test_views"], r["scoring"], r["tokens_per_image_mean"], r["setup_s"], r["eval_s"], r["poses_per_s"], r["sweep_tflops"])
PY
