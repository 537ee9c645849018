# Round 6, call 32: k_topk_small with the unordered gather: top-k / select tests, kernel durations in the headline step.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06c32; mkdir -p $O
export SIXDGS_RANDOM_BACKBONE=1
cd $R
( time timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_select.py -q -x 2>&1 | tail -n 8 ) > $O/tests.log 2>&1
grep -E "passed|failed|Error|assert " $O/tests.log | head
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $O/t -o trace -- python $R/bench.py --steps 6 --warmup 2 --skip-cpu-baseline --skip-reference-mode --l32-steps 0 --b8-steps 0 > $O/bench.json 2> $O/bench.err
DB=$(find $O/t -name "*.db" | head -1)
python $R/tools/rocpd_summary.py $DB 2>&1 | grep -E "k_topk_small|k_sel_finish|k_solve_pose|k_logits_f16x<0, 3"
python $R/tools/rocpd_timeline.py $DB $O/timeline.md "k_logits_f16x<0, 3" 2 > /dev/null 2>&1
rm -rf $O/t
head -c 300 $O/bench.json; echo
