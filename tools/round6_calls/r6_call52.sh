# Round 6, call 52: cfg-2 (one tile per launch) with the one-term pre-pass against three terms: four alternating runs + the kernels' durations under rocprofv3.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06c52; mkdir -p $O
export SIXDGS_RANDOM_BACKBONE=1
cd $R
( timeout 600 python -m pytest tests/test_gpu_select.py -q -x -k "prepass_terms or grouping" 2>&1 | tail -n 3 ) > $O/tests.log 2>&1; grep -E "passed|failed|Error|assert " $O/tests.log | head
for rep in 1 2 3 4; do for v in 3 1; do
  SIXDGS_PREPASS_TERMS=$v python -W ignore bench.py --config cfg2 --steps 60 --warmup 3 --skip-cpu-baseline --skip-reference-mode --l32-steps 0 > $O/bench_t${v}_$rep.json 2> $O/bench_t${v}_$rep.err
  python -c "
import json;d=json.loads([l for l in open('$O/bench_t${v}_$rep.json') if l.startswith('{')][-1]);print('terms $v run $rep:',d['value'],d['ms_per_step'],d['median_step']['ms'],d['roofline']['avg_launch_ms'])"
done; done
cd /tmp && export TMPDIR=/tmp
for v in 3 1; do
  SIXDGS_PREPASS_TERMS=$v timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace$v -o trace -- python $R/bench.py --config cfg2 --steps 30 --warmup 2 --skip-cpu-baseline --skip-reference-mode --l32-steps 0 > $O/trace$v.json 2> $O/trace$v.err
  DB=$(find $O/trace$v -name "*.db" | head -1); python $R/tools/rocpd_summary.py $DB > $O/trace${v}_summary.md 2>&1; echo "== terms $v"; grep "k_logits\|k_sel_finish\|k_topk_small" $O/trace${v}_summary.md | head -6
  rm -rf $O/trace$v
done
