# Round 6, call 10: k_tok_gemm (packed weight planes, token tile staged once): backbone tests, per-stage GPU time inside graphs, kernel trace at 1 / 16 images.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06c10; mkdir -p $O
export SIXDGS_RANDOM_BACKBONE=1
cd $R
( time python -m pytest tests/test_gpu_backbone.py -q -x 2>&1 | tail -n 25 ) > $O/backbone_tests.log 2>&1
tail -n 12 $O/backbone_tests.log
timeout 600 python -W ignore tools/time_vit_gemms.py > $O/vit_stages.md 2> $O/vit_stages.err
cat $O/vit_stages.md; tail -n 5 $O/vit_stages.err
cd /tmp && export TMPDIR=/tmp
for im in 1 16; do
  SIXDGS_VIT_FUSED=1 timeout 300 rocprofv3 --kernel-trace --stats -d $O/t_${im} -o trace -- python $R/tools/trace_vit.py $im 40 > $O/run_${im}.log 2>&1
  DB=$(find $O/t_${im} -name "*.db" | head -1)
  python $R/tools/rocpd_summary.py $DB > $O/summary_${im}_fused.md 2>&1
  rm -rf $O/t_${im}
  tail -n 1 $O/run_${im}.log; head -n 12 $O/summary_${im}_fused.md
done
