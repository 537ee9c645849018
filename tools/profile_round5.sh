# Round-5 evidence: rocprofv3 kernel trace of the default bench command + one PMC pass per counter (never combined with other trace
# domains), for the dominant kernel and the ray-MLP chain.   bash tools/profile_round4.sh <out-name>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r05p}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $O/trace -o trace -- python $R/bench.py --steps 5 --warmup 1 --skip-cpu-baseline --skip-reference-mode --l32-steps 0 > $O/trace_bench.json 2> $O/trace.err
DB=$(find $O/trace -name "*.db" | head -1)
python $R/tools/rocpd_summary.py $DB > $O/trace_summary.md 2>&1; head -12 $O/trace_summary.md
rm -rf $O/trace
for C in FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES; do
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv --kernel-include-regex "k_logits|k_sel_finish|k_dense_planes" -d $O/pmc_$C -o pmc -- python $R/bench.py --steps 1 --warmup 0 --skip-cpu-baseline --skip-reference-mode --l32-steps 0 > $O/pmc_$C.json 2> $O/pmc_$C.err
  python $R/tools/pmc_summary.py $O/pmc_$C | tee -a $O/pmc_raw.txt
  rm -rf $O/pmc_$C
done
# the chain alone, layer by layer (launch order within each chunk: layers 1..5), bytes per ray
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv --kernel-include-regex "k_dense_planes|k_ray_encode_planes" -d $O/pmck_$C -o pmc -- python $R/tools/time_keys.py 1048576 > /dev/null 2> $O/pmck_$C.err
  python - <<PY | tee -a $O/pmc_chain.txt
import csv, glob
rows=[]
for p in glob.glob("$O/pmck_$C/**/*counter_collection.csv", recursive=True):
    rows += [r for r in csv.DictReader(open(p)) if "k_dense_planes" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
vals=[float(r["Counter_Value"]) for r in rows]
NL=4  # round 5: four launches per chunk (k_proj folded into layer 4)
n=len(vals)//NL
for l in range(NL):
    v=vals[l::NL][n//2:]
    print("$C layer", l+1, "mean raw KB per launch of 1048576 rays (tools/time_keys.py 1048576: one chunk)", sum(v)/len(v), "launches", len(v))
PY
  rm -rf $O/pmck_$C
done
du -sh $O
