# Round 4, sixth GPU call: k_topk_small with wave-aggregated histograms (parity + cfg2 timing); images per sweep launch: where does the L2 stop holding the q planes?
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r4c6; mkdir -p $O
timeout 600 python -m pytest -q -m gpu -p no:cacheprovider -x tests/test_gpu_parity.py -k "topk" 2>&1 | tail -3
timeout 600 python -m pytest -q -m gpu -p no:cacheprovider -x tests/test_gpu_select.py 2>&1 | tail -3
for B in 4 8 12 16 24 32; do
  for cap in 0 4 8 16; do
    [ $cap -ge $B ] && [ $cap -ne 0 ] && continue
    SIXDGS_SWEEP_MAX_IMAGES=$cap timeout 200 python tools/time_sweep.py $B 8388608 3 2>&1 | grep "^sweep" | sed "s/^/cap=$cap /" >> $O/sweep_images_per_launch.log
  done
done
cat $O/sweep_images_per_launch.log | cut -c1-170
(timeout 300 python bench.py --config cfg2 --steps 20 --warmup 3 --skip-cpu-baseline --skip-reference-mode --l32-steps 0 > $O/bench_cfg2.json 2> $O/bench_cfg2.err); python -c "import json;d=json.load(open('$O/bench_cfg2.json'));print('cfg2', d['value'], d['ms_per_step'], d['median_step'], d['roofline']['avg_launch_ms'])"
