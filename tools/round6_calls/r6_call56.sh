# Round 6, call 56: kernel timeline of one pipelined step of cfg-2 (one query per step) on the final tree.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06c56; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $O/trace -o trace -- python $R/bench.py --config cfg2 --steps 12 --warmup 3 --skip-cpu-baseline --skip-reference-mode --l32-steps 0 > $O/bench.json 2> $O/trace.err
DB=$(find $O/trace -name "*.db" | head -1)
python $R/tools/rocpd_timeline.py $DB $O/timeline.md k_solve_pose 3
rm -rf $O/trace
cut -c1-150 $O/timeline.md | head -200
