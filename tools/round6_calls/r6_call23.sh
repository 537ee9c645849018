# Round 6, call 23: sixdgs_tok_attention: backbone tests, stage table (attention row), image side as replayed.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06c23; mkdir -p $O
export SIXDGS_RANDOM_BACKBONE=1
cd $R
( time timeout 600 python -m pytest tests/test_gpu_backbone.py -q -x 2>&1 | tail -n 25 ) > $O/backbone_tests.log 2>&1
grep -E "passed|failed|Error|assert " $O/backbone_tests.log | head
timeout 900 python -W ignore tools/time_vit_gemms.py > $O/vit_stages.md 2> $O/vit_stages.err
grep -E "attention|^\| images|forward|^\| [0-9]+ \| [0-9]+ \| [0-9]+ \|" $O/vit_stages.md; tail -n 5 $O/vit_stages.err
python -W ignore tools/time_image_side.py 2>&1 | grep -v amdgpu.ids | tee $O/image_side.md
