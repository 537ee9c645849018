# Round 6, call 35: k_topk_small (re-applied) -- parity + select tests.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06c35; mkdir -p $O
export SIXDGS_RANDOM_BACKBONE=1
cd $R
( time timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_select.py tests/test_gpu_e2e.py -q -x 2>&1 | tail -n 5 ) > $O/tests.log 2>&1
grep -E "passed|failed|Error|assert " $O/tests.log | head
python -W ignore bench.py --steps 10 --warmup 2 --skip-cpu-baseline --l32-steps 0 > $O/bench.json 2> $O/bench.err; head -c 250 $O/bench.json; echo
