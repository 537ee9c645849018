# Round-6 evidence: rocprofv3 kernel trace of the default bench command + one PMC pass per counter (never combined with other trace
# domains) for the dominant kernel.   bash tools/profile_round6.sh <out-name>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r06p}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $O/trace -o trace -- python $R/bench.py --steps 5 --warmup 1 --skip-cpu-baseline --skip-reference-mode --l32-steps 0 --b8-steps 0 > $O/trace_bench.json 2> $O/trace.err
DB=$(find $O/trace -name "*.db" | head -1)
python $R/tools/rocpd_summary.py $DB > $O/trace_summary.md 2>&1; head -12 $O/trace_summary.md
rm -rf $O/trace
for C in FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES; do
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv --kernel-include-regex "k_logits|k_sel_finish|k_dense_planes" -d $O/pmc_$C -o pmc -- python $R/bench.py --steps 1 --warmup 0 --skip-cpu-baseline --skip-reference-mode --l32-steps 0 --b8-steps 0 > $O/pmc_$C.json 2> $O/pmc_$C.err
  python $R/tools/pmc_summary.py $O/pmc_$C | tee -a $O/pmc_raw.txt
  rm -rf $O/pmc_$C
done
du -sh $O
