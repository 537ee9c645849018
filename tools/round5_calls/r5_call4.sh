# Round 5, call 4: pipelined steps (PoseStream), 8-rank rehearsals on the one GPU, mid-scale cfg5 stand-in with itemised set-up, allocator probe.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05c4; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_bench_contract.py -x -q -m gpu --durations=10 -p no:cacheprovider -k "pipelined or eight or mid_scale or json_contract" > $O/new_tests.log 2>&1; grep -v "^E    +" $O/new_tests.log | tail -25
for M in "" "--no-pipeline"; do
  (timeout 300 python bench.py --skip-reference-mode --skip-cpu-baseline --l32-steps 0 --steps 10 $M > $O/bench_default$M.json 2> $O/bench_default$M.err)
  python -c "
import json;d=json.load(open('$O/bench_default$M.json'));print('$M', d['value'],d['ms_per_step'],d['median_step'],d['roofline']['avg_launch_ms'],d['scene_setup_s'])"
done
python tools/probe_expandable.py > $O/alloc_probe_default.log 2>&1; tail -8 $O/alloc_probe_default.log
PYTORCH_HIP_ALLOC_CONF=expandable_segments:True python tools/probe_expandable.py > $O/alloc_probe_expandable.log 2>&1; tail -8 $O/alloc_probe_expandable.log
