# Round 6, call 8: sixdgs_tok_linear + the five-launch ViT blocks: tests, timing per shape and per forward, the presets that feel the image side.
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r06c8; mkdir -p $O
export SIXDGS_RANDOM_BACKBONE=1
( time python -m pytest tests/test_gpu_backbone.py -q -x 2>&1 | tail -n 25 ) > $O/backbone_tests.log 2>&1
tail -n 12 $O/backbone_tests.log
python -W ignore tools/time_vit_gemms.py > $O/vit_gemms.md 2> $O/vit_gemms.err
cat $O/vit_gemms.md; tail -n 5 $O/vit_gemms.err
for v in 0 1; do
  SIXDGS_VIT_FUSED=$v python -W ignore bench.py --config cfg2 --steps 30 --skip-cpu-baseline > $O/bench_cfg2_fused$v.json 2> $O/bench_cfg2_fused$v.err
  SIXDGS_VIT_FUSED=$v python -W ignore bench.py --mode reference --batch 16 --steps 20 --skip-cpu-baseline > $O/bench_refmode16_fused$v.json 2> $O/bench_refmode16_fused$v.err
done
python - <<PY
import json
for n in ("cfg2_fused0","cfg2_fused1","refmode16_fused0","refmode16_fused1"):
    try:
        d=json.loads([l for l in open("$O/bench_"+n+".json") if l.startswith("{")][-1])
        print(n, d["value"], d["ms_per_step"], d["median_step"]["ms"], (d.get("reference_mode") or {}).get("value"))
    except Exception as e: print(n, "failed", e, open("$O/bench_"+n+".err").read()[-600:])
PY
