# Round 6, call 7: the ViT's GEMM shapes, library vs own; kernel trace + PMC passes of the bench command for profiles/.
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r06c7; mkdir -p $O
export SIXDGS_RANDOM_BACKBONE=1
python -W ignore tools/time_vit_gemms.py > $O/vit_gemms.md 2> $O/vit_gemms.err
cat $O/vit_gemms.md
python -W ignore bench.py --mode reference --steps 10 --skip-cpu-baseline > $O/bench_reference_mode.json 2> $O/bench_reference_mode.err
head -c 600 $O/bench_reference_mode.json; echo
bash tools/profile_round6.sh r06c7/prof > $O/profile.log 2>&1
tail -n 30 $O/profile.log
