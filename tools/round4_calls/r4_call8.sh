# diagnostic: why did the streamed 'stump' scene take 8.4 s after 'kitchen' in call 7 (4.6 s in call 5)?
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r4c8; mkdir -p $O
for sc in "stump" "kitchen,stump"; do
  (timeout 400 python bench.py --config cfg5-standin --scenes $sc --skip-cpu-baseline > $O/b_$sc.json 2> $O/b_$sc.err); python - <<PY
import json
d=json.load(open("$O/b_$sc.json"))
for r in d["scenes"]: print("$sc |", r["scene"], r["scoring"], r["images_per_step"], r["setup_s"], r["eval_s"], r["step_s"], r["gib_allocated_reserved_free_before_eval"], r["sweep_tflops"])
PY
done
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $O/trace -o trace -- python $GRAFT_REPO_ROOT/bench.py --config cfg5-standin --scenes kitchen,stump --skip-cpu-baseline > $O/b_traced.json 2> $O/trace.err
DB=$(find $O/trace -name "*.db" | head -1); python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $DB | head -14 | cut -c1-150; rm -rf $O/trace
python - <<PY
import json
d=json.load(open("$O/b_traced.json"))
for r in d["scenes"]: print("traced |", r["scene"], r["eval_s"], r["step_s"], r["gib_allocated_reserved_free_before_eval"])
PY
