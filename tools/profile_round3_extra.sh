R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03z; mkdir -p $O
bash $R/tools/trace_bench.sh r03z_cfg2 --config cfg2 --steps 20 --warmup 3 --skip-cpu-baseline --skip-reference-mode --l32-steps 0 > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
for C in SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS; do
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv --kernel-include-regex "k_logits_f16x|k_dense_planes" -d $O/pmc_$C -o pmc -- python $R/bench.py --steps 1 --warmup 0 --skip-cpu-baseline --skip-reference-mode --l32-steps 0 > /dev/null 2>&1
  python $R/tools/pmc_summary.py $O/pmc_$C | tee -a $O/pmc_extra.txt
  rm -rf $O/pmc_$C
done
head -30 $R/gpurun_out/r03z_cfg2/kernels.md
