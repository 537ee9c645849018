# Round 6, call 13: k_tok_gemm with per-stage dispatch: backbone tests, stage table at 1..64 images, then the presets that feel the image side, fused off / on.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06c13; mkdir -p $O
export SIXDGS_RANDOM_BACKBONE=1
cd $R
( time python -m pytest tests/test_gpu_backbone.py -q -x 2>&1 | tail -n 25 ) > $O/backbone_tests.log 2>&1
grep -E "passed|failed|Error|assert " $O/backbone_tests.log | head
timeout 900 python -W ignore tools/time_vit_gemms.py > $O/vit_stages.md 2> $O/vit_stages.err
cat $O/vit_stages.md; tail -n 5 $O/vit_stages.err
for v in 0 1; do
  SIXDGS_VIT_FUSED=$v python -W ignore bench.py --config cfg2 --steps 30 --skip-cpu-baseline > $O/bench_cfg2_fused$v.json 2> $O/bench_cfg2_fused$v.err
  SIXDGS_VIT_FUSED=$v python -W ignore bench.py --mode reference --batch 16 --steps 20 --skip-cpu-baseline > $O/bench_refmode16_fused$v.json 2> $O/bench_refmode16_fused$v.err
  SIXDGS_VIT_FUSED=$v python -W ignore bench.py --steps 10 --warmup 2 --skip-cpu-baseline --skip-reference-mode --l32-steps 0 > $O/bench_headline_fused$v.json 2> $O/bench_headline_fused$v.err
done
python - <<PY
import json
for n in ("cfg2_fused0","cfg2_fused1","refmode16_fused0","refmode16_fused1","headline_fused0","headline_fused1"):
    try:
        d=json.loads([l for l in open("$O/bench_"+n+".json") if l.startswith("{")][-1])
        print(n, d["value"], d["ms_per_step"], d["median_step"]["ms"], (d.get("headline_b8") or {}).get("value"))
    except Exception as e: print(n, "failed", e, open("$O/bench_"+n+".err").read()[-600:])
PY
