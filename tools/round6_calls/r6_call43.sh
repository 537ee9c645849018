# Round 6, call 43: images per GPU per step on the headline scene -- 4 / 8 / 16, alternating, one box.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06c43; mkdir -p $O
export SIXDGS_RANDOM_BACKBONE=1
cd $R
for rep in 1 2; do for b in 4 8 16; do
  st=$((80 / b))
  python -W ignore bench.py --batch $b --steps $st --warmup 2 --skip-cpu-baseline --skip-reference-mode --l32-steps 0 --b8-steps 0 > $O/bench_b${b}_$rep.json 2> $O/bench_b${b}_$rep.err
  python -c "
import json;d=json.loads([l for l in open('$O/bench_b${b}_$rep.json') if l.startswith('{')][-1]);print('batch $b run $rep:',d['value'],d['ms_per_step'],d['median_step']['ms'],d['roofline']['avg_launch_ms'],d['roofline']['frac'],d['config']['select_sweep_launches'])" || tail -5 $O/bench_b${b}_$rep.err
done; done
