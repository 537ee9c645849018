# Round 6, call 49: kernel timeline of one pipelined step of the default command (8 images per step) on the final tree.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06c49; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $O/trace -o trace -- python $R/bench.py --steps 6 --warmup 2 --skip-cpu-baseline --skip-reference-mode --l32-steps 0 --b8-steps 0 > $O/bench.json 2> $O/trace.err
DB=$(find $O/trace -name "*.db" | head -1)
python $R/tools/rocpd_timeline.py $DB $O/timeline.md k_solve_pose 2
rm -rf $O/trace
grep -v "Cijk\|at::native\|attn_fwd\|k_tok_\|k_im2col\|k_u8" $O/timeline.md | cut -c1-150 | head -90
