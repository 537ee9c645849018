set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r2p}; mkdir -p $O
cd $R
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o trace -- python $R/bench.py --steps 5 --warmup 1 --skip-cpu-baseline --l32-steps 0 > $O/trace_bench.json 2> $O/trace.err
DB=$(find $O/trace -name "*.db" | head -1); echo DB=$DB
python $R/tools/rocpd_summary.py $DB > $O/trace_summary.md 2>&1; head -30 $O/trace_summary.md
for C in FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE; do
  timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv --kernel-include-regex "k_logits|k_sel_finish|k_dense_planes" -d $O/pmc_$C -o pmc -- python $R/bench.py --steps 1 --warmup 0 --skip-cpu-baseline --l32-steps 0 > $O/pmc_$C.json 2> $O/pmc_$C.err
  python $R/tools/pmc_summary.py $O/pmc_$C | tee $O/pmc_$C.txt
done
rm -rf $O/trace/*/*.db $O/pmc_*/ 2>/dev/null; du -sh $O
