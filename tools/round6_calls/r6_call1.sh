# Round 6, call 1: the new full-scale cfg-5 test + the arena / refused-image test; where cfg-2's 3.7 s "key_plane_buffer_alloc" comes from; baseline lines of this box.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06c1; mkdir -p $O
export SIXDGS_RANDOM_BACKBONE=1
( time python -m pytest tests/test_gpu_select.py -q -x -k "arena or falls_back" 2>&1 | tail -5 ) > $O/select_arena.log 2>&1
( time python -m pytest tests/test_gpu_bench_contract.py -q -x -s -k "full_scale" 2>&1 | tail -15 ) > $O/cfg5_full.log 2>&1
for st in 0 1 2 3 4; do python -W ignore tools/probe_alloc.py --stage $st; done > $O/probe_alloc.log 2>&1
python -W ignore tools/probe_alloc.py --stage 0 --raw >> $O/probe_alloc.log 2>&1
python -W ignore tools/probe_alloc.py --stage 4 --raw >> $O/probe_alloc.log 2>&1
python -W ignore tools/probe_alloc.py --stage 0 --gb 49 >> $O/probe_alloc.log 2>&1
python -W ignore bench.py --config cfg2 --steps 20 --skip-cpu-baseline > $O/bench_cfg2_a.json 2> $O/bench_cfg2_a.err
python -W ignore bench.py --config cfg2 --steps 20 --skip-cpu-baseline > $O/bench_cfg2_b.json 2> $O/bench_cfg2_b.err
python -W ignore bench.py --steps 20 --warmup 2 > $O/bench_default.json 2> $O/bench_default.err
tail -3 $O/select_arena.log $O/cfg5_full.log; cat $O/probe_alloc.log
python - <<PY
import json
for f in ("bench_cfg2_a","bench_cfg2_b","bench_default"):
    try:
        d=json.loads([l for l in open("$O/"+f+".json") if l.startswith("{")][-1])
        print(f, d["value"], d["ms_per_step"], d["roofline"].get("avg_launch_ms"), d["scene_setup_s"]["total"], d["scene_setup_s"]["breakdown"])
    except Exception as e: print(f, "failed", e)
PY
