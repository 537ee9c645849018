# Round 6, call 48: non-temporal activation loads and / or output stores in the ray-MLP chain (k_dense_planes), alternating; FETCH_SIZE / WRITE_SIZE of base and act-nt.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06c48; mkdir -p $O
cd $R
for rep in 1 2 3; do for v in base actnt outnt actoutnt; do
  L=""; [ $v != base ] && L=$R/build/variants/lib_$v.so
  echo "== $v run $rep"; SIXDGS_LIB=$L python -W ignore tools/time_keys.py 8388608 2>&1 | grep "planes only"
done; done | tee $O/chain_nt_ab.log
cd /tmp && export TMPDIR=/tmp
for v in base actnt; do for C in FETCH_SIZE WRITE_SIZE; do
  L=""; [ $v != base ] && L=$R/build/variants/lib_$v.so
  SIXDGS_LIB=$L timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv --kernel-include-regex "k_dense_planes" -d $O/pmc_$v -o pmc -- python -W ignore $R/tools/time_keys.py 2097152 > $O/pmc_$v.log 2>&1
  echo "== $v" | tee -a $O/pmc_raw.txt; python $R/tools/pmc_summary.py $O/pmc_$v | tee -a $O/pmc_raw.txt
  rm -rf $O/pmc_$v
done; done
