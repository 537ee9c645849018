# Round 6, call 44: the headline at 8 images per GPU and step by default (headline_b4 beside it): contract tests, the default line, kernel trace + PMC passes of the new default command.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06c44; mkdir -p $O
export SIXDGS_RANDOM_BACKBONE=1
cd $R
( timeout 1500 python -m pytest tests/test_gpu_bench_contract.py tests/test_gpu_rccl_single.py -q -x -k "single_gpu or two_ranks or pipelined or plain_start or one_rank_gives" 2>&1 | tail -n 6 ) > $O/tests.log 2>&1
grep -E "passed|failed|Error|assert " $O/tests.log | head
(timeout 500 python bench.py > $O/bench_default.json 2> $O/bench_default.err); python -c "
import json;d=json.load(open('$O/bench_default.json'));print('default',d['value'],d['ms_per_step'],d['median_step']['ms'],d['roofline']['frac'],d['roofline']['avg_launch_ms'],d['config']['select_sweep_launches'],d['parity_vs_oracle']['top100_identical'],d['headline_b4']['value'],d['two_pass_mode']['value'],d['fp32_logits_mode']['value'],d['reference_mode']['value'])" || tail -5 $O/bench_default.err
bash tools/profile_round6.sh r06c44
