# Round 6, call 9: kernel trace of the ViT forward as a hipGraph, PyTorch's kernels vs the five-launch blocks, 1 and 16 images (GPU durations per kernel:
# the per-shape table of call 8 is bound by the host's launch rate).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06c9; mkdir -p $O
export SIXDGS_RANDOM_BACKBONE=1
cd /tmp && export TMPDIR=/tmp
for im in 1 16; do for f in 0 1; do
  SIXDGS_VIT_FUSED=$f timeout 300 rocprofv3 --kernel-trace --stats -d $O/t_${im}_$f -o trace -- python $R/tools/trace_vit.py $im 40 > $O/run_${im}_$f.log 2>&1
  DB=$(find $O/t_${im}_$f -name "*.db" | head -1)
  python $R/tools/rocpd_summary.py $DB > $O/summary_${im}_$f.md 2>&1
  rm -rf $O/t_${im}_$f
  tail -n 1 $O/run_${im}_$f.log; head -n 28 $O/summary_${im}_$f.md
done; done
