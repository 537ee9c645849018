# Round 4, last GPU call: the select / launch-grouping tests on the final tree (sweep_plan.h refactor), the cfg5 stand-in at scale, the default bench line.
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r4c10; mkdir -p $O
timeout 600 python -m pytest -q -m gpu -p no:cacheprovider tests/test_gpu_select.py tests/test_gpu_integration_stub.py 2>&1 | tail -3
(timeout 900 python bench.py --config cfg5-standin > $O/bench_cfg5_standin.json 2> $O/bench_cfg5_standin.err); tail -2 $O/bench_cfg5_standin.err | cut -c1-300; python - <<PY
import json
d=json.load(open("$O/bench_cfg5_standin.json"))
print("cfg5", d["value"], d["value_including_scene_setup"], d["parity_summary"], d["roofline"]["frac"], d["roofline"]["ray_mlp_chain_tflops"])
for r in d["scenes"]: print(r["scene"], r["rays"], r["test_views"], r["scoring"], r["tokens_per_image_mean"], r["setup_s"], r["eval_s"], r["poses_per_s"], r["sweep_tflops"], r["step_s"])
PY
(timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err); python -c "
import json;d=json.load(open('$O/bench_default.json'));print('default',d['value'],d['ms_per_step'],d['roofline']['frac'],d['roofline']['avg_launch_ms'],d['config'].get('tokens_per_image'))"
