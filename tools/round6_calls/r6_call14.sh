# Round 6, call 14: what a reference-mode step of 16 images is made of (kernel trace + timeline of one step).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06c14; mkdir -p $O
export SIXDGS_RANDOM_BACKBONE=1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/t -o trace -- python $R/bench.py --mode reference --batch 16 --steps 10 --warmup 2 --skip-cpu-baseline --l32-steps 0 > $O/bench.json 2> $O/bench.err
DB=$(find $O/t -name "*.db" | head -1)
python $R/tools/rocpd_summary.py $DB > $O/summary.md 2>&1
python $R/tools/rocpd_timeline.py $DB > $O/timeline.md 2>&1
python $R/tools/step_categories.py $DB > $O/categories.md 2>&1
rm -rf $O/t
head -n 45 $O/summary.md; head -n 60 $O/categories.md
