# Round 6, call 31: kernel timeline of one pipelined headline step on the final tree (what sits between two sweeps now).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06c31; mkdir -p $O
export SIXDGS_RANDOM_BACKBONE=1
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $O/t -o trace -- python $R/bench.py --steps 6 --warmup 2 --skip-cpu-baseline --skip-reference-mode --l32-steps 0 --b8-steps 0 > $O/bench.json 2> $O/bench.err
DB=$(find $O/t -name "*.db" | head -1)
python $R/tools/rocpd_timeline.py $DB $O/timeline.md k_logits_f16x\<0,\ 3 2 > /dev/null 2>&1 || python $R/tools/rocpd_timeline.py $DB > $O/timeline.md 2>&1
rm -rf $O/t
head -c 400 $O/bench.json; echo; wc -l $O/timeline.md
