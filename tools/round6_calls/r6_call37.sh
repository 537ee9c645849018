# Round 6, call 37: timeline of a headline step with the image side one batch further ahead (lead); poses of lead / default / --no-pipeline.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06c37; mkdir -p $O
export SIXDGS_RANDOM_BACKBONE=1
cd /tmp && export TMPDIR=/tmp
SIXDGS_POSE_STREAM_LEAD=1 timeout 400 rocprofv3 --kernel-trace --stats -d $O/t -o trace -- python $R/bench.py --steps 8 --warmup 2 --skip-cpu-baseline --skip-reference-mode --l32-steps 0 --b8-steps 0 > $O/bench.json 2> $O/bench.err
DB=$(find $O/t -name "*.db" | head -1)
python $R/tools/rocpd_timeline.py $DB $O/timeline.md "k_logits_f16x<0, 3" 3 > /dev/null 2>&1
rm -rf $O/t
cd $R
for v in lead nolead nopipe; do
  if [ $v = lead ]; then export SIXDGS_POSE_STREAM_LEAD=1; else export SIXDGS_POSE_STREAM_LEAD=0; fi
  X=""; if [ $v = nopipe ]; then X="--no-pipeline"; fi
  SIXDGS_BENCH_DUMP_POSES=1 python -W ignore bench.py --steps 3 --warmup 1 $X --skip-cpu-baseline --skip-reference-mode --l32-steps 0 --b8-steps 0 > $O/poses_$v.json 2> $O/poses_$v.err
done
python - <<PY
import json
g = lambda v: json.loads([l for l in open("$O/poses_%s.json" % v) if l.startswith("{")][-1])["poses_last_step"]
ref = g("nopipe")
for v in ("lead", "nolead"): print(v, "poses identical to --no-pipeline:", g(v) == ref)
PY
