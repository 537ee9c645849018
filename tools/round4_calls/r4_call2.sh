# Round 4, second GPU call: GPU suite on the round's changes, token-aware sweep timing, ablation power table, cfg2 timelines (eager / graph), cfg5 stand-in.
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r4c2; mkdir -p $O
(timeout 1500 python -m pytest tests -q -m gpu --durations=10 -p no:cacheprovider 2>&1 | grep -v "^E    +" | tail -60) > $O/gpu_suite.log 2>&1
tail -25 $O/gpu_suite.log
for t in 256 192 128 64; do timeout 200 python tools/time_sweep.py 4 32000000 5 --tokens $t 2>&1 | grep "^sweep" >> $O/time_sweep_tokens.log; done
timeout 200 python tools/time_sweep.py 1 19200000 5 --tokens 256 2>&1 | grep "^sweep" >> $O/time_sweep_tokens.log
timeout 200 python tools/time_sweep.py 1 19200000 5 --tokens 128 2>&1 | grep "^sweep" >> $O/time_sweep_tokens.log
cat $O/time_sweep_tokens.log
for a in abl0 abl2 abl18 abl11 abl27 abl59; do
  SIXDGS_LIB=$PWD/build/variants/lib_abl.so POWER_RAYS=16000000 timeout 150 python tools/power_trace.py $a 7 2>&1 | grep -E "^POWER" >> $O/power_ablation.log
done
cat $O/power_ablation.log | cut -c1-420
cd /tmp && export TMPDIR=/tmp
for g in eager graph; do
  GA=""; [ $g = graph ] && GA="--graph"
  timeout 300 rocprofv3 --kernel-trace -d $O/trace_$g -o trace -- python $GRAFT_REPO_ROOT/bench.py --config cfg2 $GA --steps 10 --warmup 2 --skip-cpu-baseline --skip-reference-mode --l32-steps 0 > $O/bench_cfg2_$g.json 2> $O/trace_$g.err
  DB=$(find $O/trace_$g -name "*.db" | head -1)
  python $GRAFT_REPO_ROOT/tools/rocpd_timeline.py $DB $O/cfg2_${g}_timeline.md 2>&1 | tail -3
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $DB $O/cfg2_${g}_kernels.md
  rm -rf $O/trace_$g
  python -c "import json;d=json.load(open('$O/bench_cfg2_$g.json'));print('cfg2 $g', d['value'], d['ms_per_step'], d['median_step'])"
done
tail -30 $O/cfg2_eager_timeline.md
cd $GRAFT_REPO_ROOT
(timeout 900 python bench.py --config cfg5-standin > $O/bench_cfg5_standin.json 2> $O/bench_cfg5_standin.err); tail -3 $O/bench_cfg5_standin.err
python - <<PY
import json
d=json.load(open("$O/bench_cfg5_standin.json"))
print("cfg5", d["value"], d["value_including_scene_setup"], d["parity_summary"], d["roofline"]["frac"], d["roofline"]["ray_mlp_chain_tflops"])
for r in d["scenes"]: print(r["scene"], r["gaussians"], r["rays"], r["test_views"], r["scoring"], r["scoring_path"], r["tokens_per_image_mean"], r["setup_s"], r["eval_s"], r["poses_per_s"], r["sweep_tflops"], {k: r.get("parity_vs_oracle",{}).get(k) for k in ("top100_identical_select","score_rel_err","pose_rel_err")})
PY
