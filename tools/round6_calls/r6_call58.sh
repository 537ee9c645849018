# Round 6, call 58: the one-slot select sweep (cfg-2) persistent on all but r CUs, r = -1 (one-shot grid) / 0 / 8 / 16 / 32: does the next query's image side fit beside it, and what does the sweep pay?
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06c58; mkdir -p $O
export SIXDGS_RANDOM_BACKBONE=1
cd $R
for rep in 1 2; do for r in -1 0 8 16 32; do
  SIXDGS_SWEEP_ONE_SLOT_RESERVE=$r SIXDGS_BENCH_DUMP_POSES=1 python -W ignore bench.py --config cfg2 --steps 60 --warmup 3 --skip-cpu-baseline --skip-reference-mode --l32-steps 0 > $O/bench_r${r}_$rep.json 2> $O/bench_r${r}_$rep.err
  python -c "
import json;d=json.loads([l for l in open('$O/bench_r${r}_$rep.json') if l.startswith('{')][-1]);print('one-slot sweep reserve $r run $rep:',d['value'],d['ms_per_step'],d['median_step']['ms'],d['roofline']['avg_launch_ms'],round(d['ms_per_step']-d['roofline']['avg_launch_ms'],3))" || tail -5 $O/bench_r${r}_$rep.err
done; done
python - <<PY
import json
g = lambda n: json.loads([l for l in open("$O/bench_%s.json" % n) if l.startswith("{")][-1])["poses_last_step"]
for a in ("r0_1", "r8_1", "r32_1"): print(a, "poses identical to the one-shot grid:", g(a) == g("r-1_1"))
PY
