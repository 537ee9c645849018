# PMC passes over tools/time_keys.py for k_dense_planes: which pipe the kernel waits on
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r2m}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u | tr '\n' ' ' > $O/sq_counters.txt
for C in SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_WAIT_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM; do
  timeout 120 rocprofv3 --pmc $C --kernel-trace --output-format csv --kernel-include-regex "k_dense_planes" -d $O/pmc_$C -o pmc -- python $R/tools/time_keys.py 1048576 > /dev/null 2> $O/pmc_$C.err
  python $R/tools/pmc_summary.py $O/pmc_$C | tee -a $O/pmc_keys.txt
  rm -rf $O/pmc_$C
done
