# Round 6, call 60: k_tok_gemm with 4 computing waves per feature tile on N = 384 layers (proj, FC2: three full tiles instead of one full + one half) against SIXDGS_TOK_WPT=8:
# backbone tests, the stage table, the image side as replayed, reference mode.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06c60; mkdir -p $O
export SIXDGS_RANDOM_BACKBONE=1
cd $R
( timeout 900 python -m pytest tests/test_gpu_backbone.py -q -x 2>&1 | tail -n 5 ) > $O/tests.log 2>&1; grep -E "passed|failed|Error|^E " $O/tests.log | head
( SIXDGS_TOK_WPT=8 timeout 900 python -m pytest tests/test_gpu_backbone.py -q -x -k "tok_linear or fused_vit" 2>&1 | tail -n 3 ) > $O/tests8.log 2>&1; grep -E "passed|failed|Error|^E " $O/tests8.log | head
for w in 0 8; do echo "== SIXDGS_TOK_WPT=$w"; SIXDGS_TOK_WPT=$w python -W ignore tools/time_vit_gemms.py 2>&1 | grep -v amdgpu.ids | tee $O/vit_stages_wpt$w.md | grep -i "proj\|fc2\|forward\|images" | head -40; done
for w in 0 8; do echo "== SIXDGS_TOK_WPT=$w"; SIXDGS_TOK_WPT=$w python -W ignore tools/time_image_side.py 2>&1 | grep -v amdgpu.ids | tee $O/image_side_wpt$w.md; done
for rep in 1 2; do for w in 0 8; do
  SIXDGS_TOK_WPT=$w python -W ignore bench.py --mode reference --batch 16 --steps 30 --skip-cpu-baseline > $O/bench_ref_w${w}_$rep.json 2> $O/bench_ref_w${w}_$rep.err
  python -c "
import json;d=json.loads([l for l in open('$O/bench_ref_w${w}_$rep.json') if l.startswith('{')][-1]);print('reference mode, wpt $w run $rep:',d['value'],d['ms_per_step'])"
done; done
