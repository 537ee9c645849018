# Round 6, call 54: one-term pre-pass for every launch + 160 CUs left out at one slot: select / backbone tests; cfg-2 at reserve 160 (default) / 192 / 224;
# reference mode and the image side with sixdgs_image_prep against SIXDGS_IMAGE_PREP=0.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06c54; mkdir -p $O
export SIXDGS_RANDOM_BACKBONE=1
cd $R
( timeout 1500 python -m pytest tests/test_gpu_backbone.py tests/test_gpu_select.py tests/test_gpu_e2e.py -q -x -s 2>&1 | grep -E "image_prep \(|passed|failed|Error|^E " | tail -n 12 ) > $O/tests.log 2>&1; cat $O/tests.log | cut -c1-1500
for rep in 1 2; do for r in -2 192 224; do
  E=""; [ $r != -2 ] && E="SIXDGS_PREPASS_RESERVE_CUS=$r"
  env $E python -W ignore bench.py --config cfg2 --steps 60 --warmup 3 --skip-cpu-baseline --skip-reference-mode --l32-steps 0 > $O/bench_cfg2_r${r}_$rep.json 2> $O/bench_cfg2_r${r}_$rep.err
  python -c "
import json;d=json.loads([l for l in open('$O/bench_cfg2_r${r}_$rep.json') if l.startswith('{')][-1]);print('cfg2 reserve $r run $rep:',d['value'],d['ms_per_step'],d['median_step']['ms'],d['roofline']['avg_launch_ms'])"
done; done
for rep in 1 2; do for p in 1 0; do
  SIXDGS_IMAGE_PREP=$p python -W ignore bench.py --mode reference --batch 16 --steps 30 --skip-cpu-baseline > $O/bench_ref_p${p}_$rep.json 2> $O/bench_ref_p${p}_$rep.err
  python -c "
import json;d=json.loads([l for l in open('$O/bench_ref_p${p}_$rep.json') if l.startswith('{')][-1]);print('reference mode, image_prep $p run $rep:',d['value'],d['ms_per_step'])"
done; done
for p in 1 0; do echo "== SIXDGS_IMAGE_PREP=$p"; SIXDGS_IMAGE_PREP=$p python -W ignore tools/time_image_side.py 2>&1 | grep -v amdgpu.ids | tee $O/image_side_p$p.md; done
