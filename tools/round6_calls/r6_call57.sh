# Round 6, call 57: PoseStream.prefetch (the next batch's image side enqueued behind the current batch's scorer launches) against SIXDGS_POSE_PREFETCH=0: cfg-2, headline, reference mode; poses against --no-pipeline.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06c57; mkdir -p $O
export SIXDGS_RANDOM_BACKBONE=1
cd $R
for rep in 1 2 3; do for v in 0 1; do
  SIXDGS_POSE_PREFETCH=$v SIXDGS_BENCH_DUMP_POSES=1 python -W ignore bench.py --config cfg2 --steps 60 --warmup 3 --skip-cpu-baseline --skip-reference-mode --l32-steps 0 > $O/bench_cfg2_p${v}_$rep.json 2> $O/bench_cfg2_p${v}_$rep.err
  python -c "
import json;d=json.loads([l for l in open('$O/bench_cfg2_p${v}_$rep.json') if l.startswith('{')][-1]);print('cfg2 prefetch $v run $rep:',d['value'],d['ms_per_step'],d['median_step']['ms'],d['roofline']['avg_launch_ms'])" || tail -5 $O/bench_cfg2_p${v}_$rep.err
done; done
for rep in 1 2; do for v in 0 1; do
  SIXDGS_POSE_PREFETCH=$v SIXDGS_BENCH_DUMP_POSES=1 python -W ignore bench.py --steps 10 --warmup 2 --skip-cpu-baseline --skip-reference-mode --l32-steps 0 --b8-steps 10 > $O/bench_head_p${v}_$rep.json 2> $O/bench_head_p${v}_$rep.err
  python -c "
import json;d=json.loads([l for l in open('$O/bench_head_p${v}_$rep.json') if l.startswith('{')][-1]);print('headline prefetch $v run $rep:',d['value'],d['ms_per_step'],d['roofline']['avg_launch_ms'],round(d['ms_per_step']-d['roofline']['avg_launch_ms'],3),'b4',d['headline_b4']['value'],d['headline_b4']['ms_per_step'])" || tail -5 $O/bench_head_p${v}_$rep.err
  SIXDGS_POSE_PREFETCH=$v python -W ignore bench.py --mode reference --batch 16 --steps 30 --skip-cpu-baseline > $O/bench_ref_p${v}_$rep.json 2> $O/bench_ref_p${v}_$rep.err
  python -c "
import json;d=json.loads([l for l in open('$O/bench_ref_p${v}_$rep.json') if l.startswith('{')][-1]);print('reference mode prefetch $v run $rep:',d['value'],d['ms_per_step'])" || tail -5 $O/bench_ref_p${v}_$rep.err
done; done
SIXDGS_BENCH_DUMP_POSES=1 python -W ignore bench.py --config cfg2 --steps 3 --warmup 1 --no-pipeline --skip-cpu-baseline --skip-reference-mode --l32-steps 0 > $O/bench_cfg2_nopipe.json 2> $O/bench_cfg2_nopipe.err
SIXDGS_BENCH_DUMP_POSES=1 python -W ignore bench.py --steps 2 --warmup 1 --no-pipeline --skip-cpu-baseline --skip-reference-mode --l32-steps 0 --b8-steps 0 > $O/bench_head_nopipe.json 2> $O/bench_head_nopipe.err
python - <<PY
import json
g = lambda n: json.loads([l for l in open("$O/bench_%s.json" % n) if l.startswith("{")][-1])["poses_last_step"]
for a, b in (("cfg2_p1_1", "cfg2_nopipe"), ("cfg2_p0_1", "cfg2_nopipe"), ("head_p1_1", "head_nopipe")): print(a, "poses identical to --no-pipeline:", g(a) == g(b))
PY
