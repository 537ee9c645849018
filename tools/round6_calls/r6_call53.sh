# Round 6, call 53: (a) sixdgs_image_prep against the transform pipeline it replaces; (b) cfg-2 with the one-term pre-pass and 64 / 96 / 128 / 160 CUs left to the image side, against three terms at 64.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06c53; mkdir -p $O
export SIXDGS_RANDOM_BACKBONE=1
cd $R
( timeout 900 python -m pytest tests/test_gpu_backbone.py -q -x 2>&1 | tail -n 25 ) > $O/tests.log 2>&1; grep -E "passed|failed|Error|assert |^E " $O/tests.log | head -20
for rep in 1 2 3; do for v in "3 64" "1 64" "1 96" "1 128" "1 160"; do
  set -- $v
  SIXDGS_PREPASS_TERMS=$1 SIXDGS_PREPASS_RESERVE_CUS=$2 python -W ignore bench.py --config cfg2 --steps 60 --warmup 3 --skip-cpu-baseline --skip-reference-mode --l32-steps 0 > $O/bench_t$1_r$2_$rep.json 2> $O/bench_t$1_r$2_$rep.err
  python -c "
import json;d=json.loads([l for l in open('$O/bench_t$1_r$2_$rep.json') if l.startswith('{')][-1]);print('terms $1 reserve $2 run $rep:',d['value'],d['ms_per_step'],d['median_step']['ms'],d['roofline']['avg_launch_ms'])"
done; done
