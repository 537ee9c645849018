# per-layer times of the plane-to-plane chain from a rocprofv3 kernel trace:  bash tools/trace_keys.sh <out-name>   (SIXDGS_LIB selects a variant)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-trace_keys}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $O/trace -o trace -- python $R/tools/time_keys.py 4194304 > $O/time_keys.log 2> $O/trace.err
DB=$(find $O/trace -name "*.db" | head -1)
python - <<PY | tee $O/layers.txt
import sqlite3
db=sqlite3.connect("$DB"); cur=db.cursor()
rows=list(cur.execute("select name, grid_x, count(*), avg(duration), sum(duration) from kernels where name like '%k_dense%' or name like '%k_ray_encode%' or name like '%k_split_tiles%' or name like '%k_linear%' group by name, grid_x order by sum(duration) desc"))
for r in rows[:20]: print(r[0][:60], r[1], r[2], round(r[3]/1e3,1),'us avg', round(r[4]/1e6,2),'ms total')
# the layers of one chunk in launch order (layer = position within each group of L k_dense_planes launches: L = 4 with k_proj folded into layer 4
# -- the default since round 5 -- 5 with SIXDGS_FOLD_KPROJ=0); only the launches of the first timing loop (planes only) are plane-chain launches
import os
L = 5 if os.environ.get("SIXDGS_FOLD_KPROJ", "1") == "0" else 4
seq=[r[0] for r in cur.execute("select duration from kernels where name like '%k_dense_planes%' order by start")]
n=len(seq)//L
for l in range(L):
    v=sorted(seq[l::L][n//2:])            # second half of the launches: warm
    print('layer', l+1, 'of', L, 'median', round(v[len(v)//2]/1e3,1), 'us per launch (2^20 rays)')
PY
rm -rf $O/trace
