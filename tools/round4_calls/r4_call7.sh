# Round 4, seventh GPU call: launches of 8 images -- grouping test, cfg-4 and the streamed stand-in scenes again.
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r4c7; mkdir -p $O
timeout 600 python -m pytest -q -m gpu -p no:cacheprovider -x tests/test_gpu_select.py tests/test_gpu_full_size.py 2>&1 | tail -3
(timeout 500 python bench.py --config cfg4 --skip-cpu-baseline --skip-reference-mode > $O/bench_cfg4.json 2> $O/bench_cfg4.err); python -c "
import json;d=json.load(open('$O/bench_cfg4.json'));print('cfg4',d['value'],d['ms_per_step'],d['config']['scoring_path'],d['roofline']['frac'],d['roofline']['avg_launch_ms'])"
(timeout 600 python bench.py --config cfg5-standin --scenes bicycle,garden,stump,kitchen --skip-cpu-baseline > $O/bench_cfg5_big.json 2> $O/bench_cfg5_big.err); python - <<PY
import json
d=json.load(open("$O/bench_cfg5_big.json"))
print("cfg5 big", d["value"], d["roofline"]["frac"])
for r in d["scenes"]: print(r["scene"], r["rays"], r["test_views"], r["scoring"], r["images_per_step"], r["setup_s"], r["eval_s"], r["poses_per_s"], r["sweep_tflops"])
PY
