# Round 5, call 19: the ViT blocks' residuals as addcmul (24 launches fewer per image-side replay): backbone / e2e tests, cfg-2 and headline.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05c19; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_backbone.py tests/test_gpu_e2e.py -q -m gpu -p no:cacheprovider > $O/tests.log 2>&1; grep -v "^E    +" $O/tests.log | tail -4
(timeout 300 python bench.py --config cfg2 --steps 20 --skip-cpu-baseline --skip-reference-mode --l32-steps 0 > $O/bench_cfg2.json 2> $O/bench_cfg2.err)
(timeout 300 python bench.py --steps 20 --warmup 5 --skip-cpu-baseline --skip-reference-mode --l32-steps 0 > $O/bench_headline.json 2> $O/bench_headline.err)
python - <<PY
import json
for n in ('cfg2','headline'):
    d=json.load(open('$O/bench_%s.json'%n)); print(n, d['value'], d['ms_per_step'], d['median_step']['ms'], d['median_step']['min_ms'], d['roofline']['avg_launch_ms'], round(d['median_step']['ms']-d['roofline']['avg_launch_ms'],3))
PY
