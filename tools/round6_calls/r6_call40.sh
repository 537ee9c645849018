# Round 6, call 40: pre-pass leaving 64 / 96 / 128 CUs (continuation of call 39).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06c40; mkdir -p $O
export SIXDGS_RANDOM_BACKBONE=1
cd $R
for rep in 1 2; do for rs in -1 64 96 128; do
  SIXDGS_PREPASS_RESERVE_CUS=$rs python -W ignore bench.py --steps 20 --warmup 2 --skip-cpu-baseline --skip-reference-mode --l32-steps 0 --b8-steps 0 > $O/bench_r${rs}_$rep.json 2> $O/bench_r${rs}_$rep.err
  python -c "
import json;d=json.loads([l for l in open('$O/bench_r${rs}_$rep.json') if l.startswith('{')][-1]);print('pre-pass leaves $rs CUs, run $rep:',d['value'],d['ms_per_step'],d['median_step']['ms'],d['roofline']['avg_launch_ms'],round(d['ms_per_step']-d['roofline']['avg_launch_ms'],3))" || tail -3 $O/bench_r${rs}_$rep.err
done; done
