# Round 5, call 9: the ViT block's attention as one kernel + addcmul residuals -- parity and what it buys (cfg-2 and the headline, fused on / off).
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05c9; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_backbone.py tests/test_gpu_e2e.py -x -q -m gpu -p no:cacheprovider > $O/vit_tests.log 2>&1; grep -v "^E    +" $O/vit_tests.log | tail -8
for F in 1 0; do
  (SIXDGS_FUSED_VIT=$F timeout 300 python bench.py --config cfg2 --steps 20 --skip-cpu-baseline --skip-reference-mode --l32-steps 0 > $O/bench_cfg2_fused$F.json 2> $O/bench_cfg2_fused$F.err)
  (SIXDGS_FUSED_VIT=$F timeout 300 python bench.py --config cfg2 --steps 20 --no-pipeline --skip-cpu-baseline --skip-reference-mode --l32-steps 0 > $O/bench_cfg2_fused${F}_nopipe.json 2> $O/bench_cfg2_fused${F}_nopipe.err)
  (SIXDGS_FUSED_VIT=$F timeout 300 python bench.py --steps 10 --skip-cpu-baseline --skip-reference-mode --l32-steps 0 > $O/bench_headline_fused$F.json 2> $O/bench_headline_fused$F.err)
  python - <<PY
import json
for n in ('cfg2_fused$F','cfg2_fused${F}_nopipe','headline_fused$F'):
    d=json.load(open('$O/bench_%s.json'%n)); print(n, d['value'], d['ms_per_step'], d['median_step']['ms'], d['median_step']['min_ms'], d['roofline']['avg_launch_ms'])
PY
done
