# Round 6, call 50: the sample pre-pass with ONE MFMA term (h*h) instead of three: select / config tests under it, A/B on the headline (8, 4 images), cfg-2, cfg-3; candidates per image.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06c50; mkdir -p $O
export SIXDGS_RANDOM_BACKBONE=1
cd $R
( SIXDGS_PREPASS_TERMS=1 timeout 1500 python -m pytest tests/test_gpu_select.py tests/test_gpu_ray_sharded.py tests/test_gpu_full_size.py -q -x 2>&1 | tail -n 4 ) > $O/tests.log 2>&1
grep -E "passed|failed|Error|assert " $O/tests.log | head
for rep in 1 2; do for v in 3 1; do for c in "headline --batch 8" "headline --batch 4" cfg2 cfg3; do
  n=$(echo $c | tr -d ' -'); st=10; [ "$c" = cfg2 ] && st=40; [ "$c" = cfg3 ] && st=5
  SIXDGS_PREPASS_TERMS=$v python -W ignore bench.py --config $c --steps $st --warmup 2 --skip-cpu-baseline --skip-reference-mode --l32-steps 0 --b8-steps 0 > $O/bench_t${v}_${n}_$rep.json 2> $O/bench_t${v}_${n}_$rep.err
  python -c "
import json;d=json.loads([l for l in open('$O/bench_t${v}_${n}_$rep.json') if l.startswith('{')][-1]);print('terms $v $n run $rep:',d['value'],d['ms_per_step'],d['median_step']['ms'],d['roofline']['avg_launch_ms'],round(d['ms_per_step']-d['roofline']['avg_launch_ms']*len(d['config']['select_sweep_launches']),3),d['config'].get('select_candidates_last_batch'))" || tail -5 $O/bench_t${v}_${n}_$rep.err
done; done; done
