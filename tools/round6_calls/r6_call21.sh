# Round 6, call 21: reference mode pipelined; bench contract tests; default bench line.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06c21; mkdir -p $O
export SIXDGS_RANDOM_BACKBONE=1
cd $R
python -W ignore bench.py --mode reference --batch 16 --steps 20 --skip-cpu-baseline > $O/bench_refmode16.json 2> $O/bench_refmode16.err
python -W ignore bench.py --mode reference --batch 16 --steps 20 --skip-cpu-baseline --no-pipeline > $O/bench_refmode16_nopipe.json 2> $O/bench_refmode16_nopipe.err
python -W ignore bench.py --mode reference --steps 20 --skip-cpu-baseline > $O/bench_refmode4.json 2> $O/bench_refmode4.err
python -W ignore bench.py --config cfg2 --steps 30 --skip-cpu-baseline > $O/bench_cfg2.json 2> $O/bench_cfg2.err
python -W ignore bench.py > $O/bench_default.json 2> $O/bench_default.err
python - <<PY
import json
for n in ("refmode16","refmode16_nopipe","refmode4","cfg2","default"):
    try:
        d=json.loads([l for l in open("$O/bench_"+n+".json") if l.startswith("{")][-1])
        print(n, d["value"], d["ms_per_step"], d["median_step"]["ms"], (d.get("headline_b8") or {}).get("value"), (d.get("reference_mode") or {}).get("value"), d["config"]["pipeline"][:30])
    except Exception as e: print(n, "failed", e, open("$O/bench_"+n+".err").read()[-800:])
PY
( time timeout 900 python -m pytest tests/test_gpu_bench_contract.py -q -x -k "not cfg5 and not eight" 2>&1 | tail -n 8 ) > $O/contract_tests.log 2>&1
tail -n 5 $O/contract_tests.log
