# rocprofv3 kernel trace of a bench.py run + per-kernel summary:  bash tools/trace_bench.sh <out-name> <bench args...>
R=$GRAFT_REPO_ROOT; N=$1; shift; O=$R/gpurun_out/$N; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace -o trace -- python $R/bench.py "$@" > $O/bench.json 2> $O/trace.err
DB=$(find $O/trace -name "*.db" | head -1)
python $R/tools/rocpd_summary.py $DB $O/kernels.md
rm -rf $O/trace
