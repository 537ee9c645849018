# Round 6, call 17: batched im2col for the camera-up CNN + one-pass uint8 -> planar fp32: tests (backbone, e2e, parity of the image side), reference mode and cfg2.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06c17; mkdir -p $O
export SIXDGS_RANDOM_BACKBONE=1
cd $R
( time python -m pytest tests/test_gpu_backbone.py tests/test_gpu_e2e.py tests/test_gpu_cfg1.py -q -x 2>&1 | tail -n 25 ) > $O/tests.log 2>&1
grep -E "passed|failed|Error|assert " $O/tests.log | head
python -W ignore bench.py --mode reference --batch 16 --steps 20 --skip-cpu-baseline > $O/bench_refmode16.json 2> $O/bench_refmode16.err
python -W ignore bench.py --config cfg2 --steps 30 --skip-cpu-baseline > $O/bench_cfg2.json 2> $O/bench_cfg2.err
python -W ignore bench.py --steps 10 --warmup 2 --skip-cpu-baseline --l32-steps 0 > $O/bench_headline.json 2> $O/bench_headline.err
python - <<PY
import json
for n in ("refmode16","cfg2","headline"):
    try:
        d=json.loads([l for l in open("$O/bench_"+n+".json") if l.startswith("{")][-1])
        print(n, d["value"], d["ms_per_step"], d["median_step"]["ms"], (d.get("headline_b8") or {}).get("value"), (d.get("reference_mode") or {}).get("value"))
    except Exception as e: print(n, "failed", e, open("$O/bench_"+n+".err").read()[-600:])
PY
