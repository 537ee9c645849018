# Round 6, call 55: the default bench line of the final tree (traffic matched on the recorded sweep plan) + kernel trace and PMC passes of the same command.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06c55; mkdir -p $O
cd $R
(timeout 500 python bench.py > $O/bench_default.json 2> $O/bench_default.err); python -c "
import json;d=json.load(open('$O/bench_default.json'));print('default',d['value'],d['ms_per_step'],d['median_step']['ms'],d['roofline']['frac'],d['roofline']['avg_launch_ms'],d['roofline']['traffic'],d['headline_b4']['value'],d['reference_mode']['value'])" || tail -5 $O/bench_default.err
bash tools/profile_round6.sh r06c55
