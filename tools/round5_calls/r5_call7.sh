# Round 5, call 7: kernel categories of one streamed step (stump: 317 M rays, 16 views in ONE step); the default bench line with the pipelined warm-up.
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05c7; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace -d $O/trace_stump -o trace -- python $R/bench.py --config cfg5-standin --scenes stump --skip-cpu-baseline > $O/trace_stump.json 2> $O/trace_stump.err
cd $R
DB=$(find $O/trace_stump -name "*.db" | head -1); python tools/step_categories.py $DB $O/streamed_step_categories.md k_solve_pose 1 k_emit_isocell; cat $O/streamed_step_categories.md; rm -rf $O/trace_stump
python -c "
import json;d=json.load(open('$O/trace_stump.json'));r=d['scenes'][0];print(r['scene'],r['rays'],r['images_per_step'],r['poses_per_s'],r['step_s'],r['sweep_tflops'])"
(timeout 300 python bench.py --skip-reference-mode --skip-cpu-baseline --l32-steps 0 --steps 10 > $O/bench_default.json 2> $O/bench_default.err)
python -c "
import json;d=json.load(open('$O/bench_default.json'));print(d['value'],d['ms_per_step'],d['median_step'],d['roofline']['avg_launch_ms'],d['roofline']['frac'])"
