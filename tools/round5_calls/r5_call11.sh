# Round 5, call 11: what the sweep at ONE tile per launch waits for -- compile-time ablations (tools/ablate_logits.py names) at B = 1 and B = 4.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05c11; mkdir -p $O
export SIXDGS_LIB=$GRAFT_REPO_ROOT/build/variants/lib_abl.so
for B in 1 4; do
  R=19200000; [ $B = 4 ] && R=32000000
  for A in 0 2 11 18 59; do
    SIXDGS_DEBUG_ABLATE=$A python tools/time_sweep.py $B $R 3 2>&1 | tail -1 | sed "s/^/abl=$A /" | tee -a $O/sweep_ablations_b1_b4.log
  done
done
