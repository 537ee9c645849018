# Round 6, call 33: the image stream at high priority (its ViT chain of the NEXT batch is stalled ~0.4 ms behind k_sel_finish_slots' 125 k workgroups): A/B, alternating.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06c33; mkdir -p $O
export SIXDGS_RANDOM_BACKBONE=1
cd $R
for rep in 1 2 3; do for pr in 0 -1; do
  SIXDGS_IMAGE_STREAM_PRIORITY=$pr python -W ignore bench.py --steps 20 --warmup 2 --skip-cpu-baseline --skip-reference-mode --l32-steps 0 --b8-steps 0 > $O/bench_p${pr}_$rep.json 2> $O/bench_p${pr}_$rep.err
  python -c "
import json;d=json.loads([l for l in open('$O/bench_p${pr}_$rep.json') if l.startswith('{')][-1]);print('priority $pr run $rep:',d['value'],d['ms_per_step'],d['median_step']['ms'],d['roofline']['avg_launch_ms'],round(d['ms_per_step']-d['roofline']['avg_launch_ms'],3))"
done; done
for pr in 0 -1; do
  SIXDGS_IMAGE_STREAM_PRIORITY=$pr python -W ignore bench.py --config cfg2 --steps 30 --skip-cpu-baseline --skip-reference-mode --l32-steps 0 > $O/bench_cfg2_p${pr}.json 2> $O/bench_cfg2_p${pr}.err
  python -c "
import json;d=json.loads([l for l in open('$O/bench_cfg2_p${pr}.json') if l.startswith('{')][-1]);print('cfg2 priority $pr:',d['value'],d['ms_per_step'],d['median_step']['ms'],d['roofline']['avg_launch_ms'])"
done
