# Round 6, call 27: k_solve_pose with per-ray terms in parallel: parity tests of the pose tail, e2e, select; kernel duration from a short trace.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06c27; mkdir -p $O
export SIXDGS_RANDOM_BACKBONE=1
cd $R
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_e2e.py tests/test_gpu_cfg1.py -q -x 2>&1 | tail -n 8 ) > $O/tests.log 2>&1
grep -E "passed|failed|Error|assert " $O/tests.log | head
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/t -o trace -- python $R/bench.py --mode reference --batch 16 --steps 10 --warmup 2 --skip-cpu-baseline --l32-steps 0 > $O/bench_ref.json 2> $O/bench_ref.err
DB=$(find $O/t -name "*.db" | head -1)
python $R/tools/rocpd_summary.py $DB 2>&1 | grep -E "k_solve_pose|k_topk_small|k_score_reduce|k_logits" 
rm -rf $O/t
timeout 300 rocprofv3 --kernel-trace --stats -d $O/t -o trace -- python $R/bench.py --steps 5 --warmup 1 --skip-cpu-baseline --skip-reference-mode --l32-steps 0 --b8-steps 0 > $O/bench_hl.json 2> $O/bench_hl.err
DB=$(find $O/t -name "*.db" | head -1)
python $R/tools/rocpd_summary.py $DB 2>&1 | grep -E "k_solve_pose|k_topk_small|k_sel_finish|k_sel_rescore"
rm -rf $O/t
