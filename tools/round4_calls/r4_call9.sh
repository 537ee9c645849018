# Round 4, final validation of the tree: smoke, whole GPU suite, bench presets (tools/validate_round4.sh), cfg5 stand-in at scale.
cd $GRAFT_REPO_ROOT
bash tools/validate_round4.sh r04v2
O=gpurun_out/r04v2
(timeout 900 python bench.py --config cfg5-standin > $O/bench_cfg5_standin.json 2> $O/bench_cfg5_standin.err); python - <<PY
import json
d=json.load(open("$O/bench_cfg5_standin.json"))
print("cfg5", d["value"], d["value_including_scene_setup"], d["parity_summary"], d["roofline"]["frac"], d["roofline"]["ray_mlp_chain_tflops"])
for r in d["scenes"]: print(r["scene"], r["rays"], r["test_views"], r["scoring"], r["tokens_per_image_mean"], r["setup_s"], r["eval_s"], r["poses_per_s"], r["sweep_tflops"], r["step_s"])
PY
