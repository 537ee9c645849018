# Round 6, call 47: non-temporal activation loads in the ray-MLP chain (k_dense_planes), base against -DSDG_ACT_NT=1, alternating.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06c47; mkdir -p $O
cd $R
for rep in 1 2 3; do for v in base actnt; do
  L=""; [ $v != base ] && L=$R/build/variants/lib_$v.so
  echo "== $v run $rep"; SIXDGS_LIB=$L python -W ignore tools/time_keys.py 8388608 2>&1 | grep "planes only"
done; done | tee $O/chain_act_nt_ab.log
