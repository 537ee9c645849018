R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03q; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for S in 0 2 1; do
  export SIXDGS_SIBLING_SYNC=$S
  python $R/bench.py --steps 10 --warmup 2 --skip-cpu-baseline --skip-reference-mode --l32-steps 0 > $O/bench_sib$S.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("$O/bench_sib$S.json")); print("sib=$S", d["value"], d["ms_per_step"], d["median_step"]["ms"], "launch", d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["config"]["select_candidates_last_batch"])
PY
  for C in FETCH_SIZE GRBM_GUI_ACTIVE; do
    timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv --kernel-include-regex "k_logits" -d $O/pmc_$C -o pmc -- python $R/bench.py --steps 1 --warmup 0 --skip-cpu-baseline --skip-reference-mode --l32-steps 0 > /dev/null 2>&1
    python $R/tools/pmc_summary.py $O/pmc_$C | grep "0, 3"
    rm -rf $O/pmc_$C
  done
done
