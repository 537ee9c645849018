# Round 6, call 59: CUs the (now one-term) pre-pass leaves out at 8 slots per launch: 32 / 64 (default) / 96 / 128, headline, alternating.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06c59; mkdir -p $O
export SIXDGS_RANDOM_BACKBONE=1
cd $R
for rep in 1 2; do for r in 64 32 96 128; do
  SIXDGS_PREPASS_RESERVE_CUS=$r python -W ignore bench.py --steps 10 --warmup 2 --skip-cpu-baseline --skip-reference-mode --l32-steps 0 --b8-steps 0 > $O/bench_r${r}_$rep.json 2> $O/bench_r${r}_$rep.err
  python -c "
import json;d=json.loads([l for l in open('$O/bench_r${r}_$rep.json') if l.startswith('{')][-1]);print('pre-pass reserve $r run $rep:',d['value'],d['ms_per_step'],d['roofline']['avg_launch_ms'],round(d['ms_per_step']-d['roofline']['avg_launch_ms'],3))" || tail -5 $O/bench_r${r}_$rep.err
done; done
