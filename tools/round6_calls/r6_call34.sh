# Round 6, call 34: persistent k_sel_finish_slots (bounded workgroups, so that the next batch's image side is not stalled behind it): select tests, A/B by workgroups per CU.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06c34; mkdir -p $O
export SIXDGS_RANDOM_BACKBONE=1
cd $R
( time timeout 1200 python -m pytest tests/test_gpu_select.py -q -x 2>&1 | tail -n 5 ) > $O/tests.log 2>&1
grep -E "passed|failed|Error|assert " $O/tests.log | head
for rep in 1 2; do for w in 1000 4 2 8; do
  SIXDGS_FINISH_WGS_PER_CU=$w python -W ignore bench.py --steps 20 --warmup 2 --skip-cpu-baseline --skip-reference-mode --l32-steps 0 --b8-steps 0 > $O/bench_w${w}_$rep.json 2> $O/bench_w${w}_$rep.err
  python -c "
import json;d=json.loads([l for l in open('$O/bench_w${w}_$rep.json') if l.startswith('{')][-1]);print('finish workgroups per CU $w run $rep:',d['value'],d['ms_per_step'],d['median_step']['ms'],d['roofline']['avg_launch_ms'],round(d['ms_per_step']-d['roofline']['avg_launch_ms'],3))"
done; done
