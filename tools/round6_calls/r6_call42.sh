# Round 6, call 42: the tail of a batch on a third stream beside the next batch's (now persistent, 192-CU) pre-pass, the next sweep waiting for it: select tests incl. the
# split entry point, A/B on the headline, poses against --no-pipeline.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06c42; mkdir -p $O
export SIXDGS_RANDOM_BACKBONE=1
cd $R
( timeout 1200 python -m pytest tests/test_gpu_select.py -q -x 2>&1 | tail -n 4 ) > $O/tests.log 2>&1
grep -E "passed|failed|Error|assert " $O/tests.log | head
for rep in 1 2 3; do for tl in 0 1; do
  SIXDGS_POSE_STREAM_TAIL=$tl SIXDGS_BENCH_DUMP_POSES=1 python -W ignore bench.py --steps 20 --warmup 2 --skip-cpu-baseline --skip-reference-mode --l32-steps 0 --b8-steps 0 > $O/bench_t${tl}_$rep.json 2> $O/bench_t${tl}_$rep.err
  python -c "
import json;d=json.loads([l for l in open('$O/bench_t${tl}_$rep.json') if l.startswith('{')][-1]);print('tail stream $tl run $rep:',d['value'],d['ms_per_step'],d['median_step']['ms'],d['roofline']['avg_launch_ms'],round(d['ms_per_step']-d['roofline']['avg_launch_ms'],3))" || tail -5 $O/bench_t${tl}_$rep.err
done; done
SIXDGS_BENCH_DUMP_POSES=1 python -W ignore bench.py --steps 3 --warmup 1 --no-pipeline --skip-cpu-baseline --skip-reference-mode --l32-steps 0 --b8-steps 0 > $O/bench_nopipe.json 2> $O/bench_nopipe.err
python - <<PY
import json
g = lambda n: json.loads([l for l in open("$O/bench_%s.json" % n) if l.startswith("{")][-1])["poses_last_step"]
ref = g("nopipe")
for n in ("t0_1", "t1_1", "t1_2"): print(n, "poses identical to --no-pipeline:", g(n) == ref)
PY
