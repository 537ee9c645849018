# Round 5, call 10: the image side as two graphs (ViT; camera-up CNN), the CNN off the path to the sweep in the pipelined step -- tests, cfg-2 and headline.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05c10; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_backbone.py tests/test_gpu_e2e.py tests/test_gpu_bench_contract.py tests/test_gpu_select.py -x -q -m gpu -p no:cacheprovider -k "not cfg5 and not eight" > $O/tests.log 2>&1; grep -v "^E    +" $O/tests.log | tail -6
(timeout 300 python bench.py --config cfg2 --steps 20 --skip-cpu-baseline --skip-reference-mode --l32-steps 0 > $O/bench_cfg2.json 2> $O/bench_cfg2.err)
(timeout 300 python bench.py --steps 10 --skip-cpu-baseline --skip-reference-mode --l32-steps 0 > $O/bench_headline.json 2> $O/bench_headline.err)
(timeout 300 python bench.py --steps 10 --no-pipeline --skip-cpu-baseline --skip-reference-mode --l32-steps 0 > $O/bench_headline_nopipe.json 2> $O/bench_headline_nopipe.err)
python - <<PY
import json
for n in ('cfg2','headline','headline_nopipe'):
    d=json.load(open('$O/bench_%s.json'%n)); print(n, d['value'], d['ms_per_step'], d['median_step']['ms'], d['median_step']['min_ms'], d['roofline']['avg_launch_ms'])
PY
