# Round 6, call 16: step timelines after the image-side changes: reference mode (16 images) and cfg2 (one image, pipelined).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06c16; mkdir -p $O
export SIXDGS_RANDOM_BACKBONE=1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/t -o trace -- python $R/bench.py --mode reference --batch 16 --steps 10 --warmup 2 --skip-cpu-baseline --l32-steps 0 > $O/bench_ref.json 2> $O/bench_ref.err
DB=$(find $O/t -name "*.db" | head -1)
python $R/tools/rocpd_timeline.py $DB > $O/timeline_ref16.md 2>&1
rm -rf $O/t
timeout 600 rocprofv3 --kernel-trace --stats -d $O/t -o trace -- python $R/bench.py --config cfg2 --steps 10 --warmup 2 --skip-cpu-baseline --skip-reference-mode --l32-steps 0 > $O/bench_cfg2.json 2> $O/bench_cfg2.err
DB=$(find $O/t -name "*.db" | head -1)
python $R/tools/rocpd_timeline.py $DB > $O/timeline_cfg2.md 2>&1
python $R/tools/rocpd_summary.py $DB > $O/summary_cfg2.md 2>&1
rm -rf $O/t
grep -v "im2col_kernel\|^| [0-9]* | .k_tok_gemm\|attn_fwd" $O/timeline_ref16.md | head -90
