# Round 6, call 36: PoseStream with the image side one batch further ahead (lead) and two handles pending in the caller's loop: A/B on the headline, poses against --no-pipeline.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06c36; mkdir -p $O
export SIXDGS_RANDOM_BACKBONE=1
cd $R
for rep in 1 2 3; do for lead in 0 1; do
  SIXDGS_POSE_STREAM_LEAD=$lead SIXDGS_BENCH_DUMP_POSES=$O/poses_lead${lead}_$rep.npy python -W ignore bench.py --steps 20 --warmup 2 --skip-cpu-baseline --skip-reference-mode --l32-steps 0 --b8-steps 0 > $O/bench_lead${lead}_$rep.json 2> $O/bench_lead${lead}_$rep.err
  python -c "
import json;d=json.loads([l for l in open('$O/bench_lead${lead}_$rep.json') if l.startswith('{')][-1]);print('lead $lead run $rep:',d['value'],d['ms_per_step'],d['median_step']['ms'],d['roofline']['avg_launch_ms'],round(d['ms_per_step']-d['roofline']['avg_launch_ms'],3))" || tail -5 $O/bench_lead${lead}_$rep.err
done; done
SIXDGS_BENCH_DUMP_POSES=$O/poses_nopipe.npy python -W ignore bench.py --steps 3 --warmup 1 --no-pipeline --skip-cpu-baseline --skip-reference-mode --l32-steps 0 --b8-steps 0 > $O/bench_nopipe.json 2> $O/bench_nopipe.err
python - <<PY
import numpy as np, glob
ref = np.load("$O/poses_nopipe.npy")
for f in sorted(glob.glob("$O/poses_lead*.npy")):
    p = np.load(f); print(f.split('/')[-1], "identical to --no-pipeline:", bool(np.array_equal(p, ref)), p.shape)
PY
