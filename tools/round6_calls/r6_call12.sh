# Round 6, call 12: where k_tok_gemm's time goes (cycle stamps of one workgroup), backbone tests, per-stage timing.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06c12; mkdir -p $O
export SIXDGS_RANDOM_BACKBONE=1
cd $R
SIXDGS_LIB=$R/build/variants/lib_tokprof.so python -W ignore tools/prof_tok.py > $O/prof_tok.log 2>&1
cat $O/prof_tok.log
( time python -m pytest tests/test_gpu_backbone.py -q -x 2>&1 | tail -n 25 ) > $O/backbone_tests.log 2>&1
grep -E "passed|failed|Error|assert " $O/backbone_tests.log | head
timeout 600 python -W ignore tools/time_vit_gemms.py > $O/vit_stages.md 2> $O/vit_stages.err
cat $O/vit_stages.md; tail -n 5 $O/vit_stages.err
