# Round 6, call 22: kernel trace + PMC passes of the default bench command on the tree with the image-side kernels (profiles/r06_*), default bench line.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06c22; mkdir -p $O
export SIXDGS_RANDOM_BACKBONE=1
cd $R
bash tools/profile_round6.sh r06c22/prof > $O/profile.log 2>&1
tail -n 40 $O/profile.log
python -W ignore bench.py > $O/bench_default.json 2> $O/bench_default.err
head -c 600 $O/bench_default.json
