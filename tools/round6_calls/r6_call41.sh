# Round 6, call 41: pre-pass leaving 64 CUs vs one-shot on the other presets: cfg2 (one image), cfg3 (8 images, 64 M rays), headline --no-pipeline, reference mode.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06c41; mkdir -p $O
export SIXDGS_RANDOM_BACKBONE=1
cd $R
for rep in 1 2; do for rs in -1 64; do
  for cfg in "cfg2:--config cfg2 --steps 30" "cfg3:--config cfg3 --steps 6" "nopipe:--no-pipeline --steps 10 --b8-steps 0" "refmode:--mode reference --batch 16 --steps 20"; do
    n=${cfg%%:*}; a=${cfg#*:}
    SIXDGS_PREPASS_RESERVE_CUS=$rs python -W ignore bench.py $a --skip-cpu-baseline --skip-reference-mode --l32-steps 0 > $O/bench_${n}_r${rs}_$rep.json 2> $O/bench_${n}_r${rs}_$rep.err
    python -c "
import json;d=json.loads([l for l in open('$O/bench_${n}_r${rs}_$rep.json') if l.startswith('{')][-1]);print('$n, pre-pass leaves $rs CUs, run $rep:',d['value'],d['ms_per_step'],d['median_step']['ms'],d['roofline'].get('avg_launch_ms'))" || tail -3 $O/bench_${n}_r${rs}_$rep.err
  done
done; done
