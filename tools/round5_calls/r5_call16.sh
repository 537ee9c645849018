# Round 5, call 16: units of 32 token rows (kOutUBH) -- the select tests, the sweep over masked batches (random token counts) with the unit forced to 4 and 8,
# the full-token headline sweep (must not move), the stand-in's Tanks&Temples scenes.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05c16; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_select.py tests/test_gpu_full_size.py -x -q -m gpu -p no:cacheprovider > $O/select_tests.log 2>&1; grep -v "^E    +" $O/select_tests.log | tail -6
for U in 4 8; do
  SIXDGS_SWEEP_UNITS=$U python tools/time_sweep.py 16 32000000 3 --tokens-range 56 180 2>&1 | tail -1 | tee -a $O/time_sweep_units.log
  SIXDGS_SWEEP_UNITS=$U python tools/time_sweep.py 32 19200000 3 --tokens-range 56 180 2>&1 | tail -1 | tee -a $O/time_sweep_units.log
done
python tools/time_sweep.py 4 32000000 5 2>&1 | tail -1 | tee -a $O/time_sweep_units.log
SIXDGS_SWEEP_UNITS=8 python tools/time_sweep.py 4 32000000 5 2>&1 | tail -1 | tee -a $O/time_sweep_units.log
(timeout 600 python bench.py --config cfg5-standin --scenes tt_ > $O/bench_cfg5_tt.json 2> $O/bench_cfg5_tt.err); python - <<PY
import json
d=json.load(open('$O/bench_cfg5_tt.json'))
print(d['value'], d['parity_summary'])
for r in d['scenes']: print(r['scene'], r['images_per_step'], r['poses_per_s'], r['sweep_tflops'], r['tokens_per_image_mean'], r['step_s'])
PY
