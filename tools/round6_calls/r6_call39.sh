# Round 6, call 39: the sample pre-pass persistent on fewer workgroups than CUs (so that the next batch's image side keeps running beside it): select tests with it on,
# A/B on the headline by CUs left out, poses against the default.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06c39; mkdir -p $O
export SIXDGS_RANDOM_BACKBONE=1
cd $R
( SIXDGS_PREPASS_RESERVE_CUS=16 timeout 1200 python -m pytest tests/test_gpu_select.py -q -x 2>&1 | tail -n 4 ) > $O/tests.log 2>&1
grep -E "passed|failed|Error|assert " $O/tests.log | head
for rep in 1 2; do for rs in -1 0 16 32 64; do
  SIXDGS_PREPASS_RESERVE_CUS=$rs SIXDGS_BENCH_DUMP_POSES=1 python -W ignore bench.py --steps 20 --warmup 2 --skip-cpu-baseline --skip-reference-mode --l32-steps 0 --b8-steps 0 > $O/bench_r${rs}_$rep.json 2> $O/bench_r${rs}_$rep.err
  python -c "
import json;d=json.loads([l for l in open('$O/bench_r${rs}_$rep.json') if l.startswith('{')][-1]);print('pre-pass leaves $rs CUs, run $rep:',d['value'],d['ms_per_step'],d['median_step']['ms'],d['roofline']['avg_launch_ms'],round(d['ms_per_step']-d['roofline']['avg_launch_ms'],3))" || tail -3 $O/bench_r${rs}_$rep.err
done; done
python - <<PY
import json
g = lambda n: json.loads([l for l in open("$O/bench_r%s_1.json" % n) if l.startswith("{")][-1])["poses_last_step"]
ref = g("-1")
for n in ("0", "16", "32", "64"): print("leaving", n, "CUs: poses identical to the one-shot pre-pass:", g(n) == ref)
PY
