# Round 5, call 17: last check of the final tree (the library was rebuilt after the reverted experiments): smoke, the select / fold / pipeline tests, the
# driver's bench command; socket power under the folded chain and the sweep.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05c17; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 900 python -m pytest tests/test_gpu_select.py tests/test_gpu_fold.py tests/test_gpu_backbone.py tests/test_gpu_e2e.py -q -m gpu -p no:cacheprovider > $O/tests.log 2>&1; grep -v "^E    +" $O/tests.log | tail -4
(timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err); python -c "
import json;d=json.load(open('$O/bench_default.json'));print('default',d['value'],d['ms_per_step'],d['median_step']['ms'],d['roofline']['frac'],d['roofline']['avg_launch_ms'],d['parity_vs_oracle']['top100_identical'],d['scene_setup_s']['ray_mlp_keys_tflops'])"
python tools/power_trace.py chain 12 2>&1 | grep POWER | tee $O/power_chain.txt
python tools/power_trace.py sweep 12 2>&1 | grep POWER | tee $O/power_sweep.txt
