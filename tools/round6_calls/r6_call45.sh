# Round 6, call 45: `nt` on the key stream's LDS-DMA loads at 8 tiles per launch (the q planes of 8 slots + the key tiles in flight overflow the XCD's L2:
# traffic 1.44 x): does the streaming hint keep the q planes resident, and does the sweep notice?  Base against variant, alternating; FETCH_SIZE of both.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06c45; mkdir -p $O
export SIXDGS_RANDOM_BACKBONE=1
cd $R
for rep in 1 2; do for v in base keynt; do for b in 8 4; do
  L=""; [ $v != base ] && L=$R/build/variants/lib_$v.so
  SIXDGS_LIB=$L python -W ignore bench.py --batch $b --steps $((80 / b)) --warmup 2 --skip-cpu-baseline --skip-reference-mode --l32-steps 0 --b8-steps 0 > $O/bench_${v}_b${b}_$rep.json 2> $O/bench_${v}_b${b}_$rep.err
  python -c "
import json;d=json.loads([l for l in open('$O/bench_${v}_b${b}_$rep.json') if l.startswith('{')][-1]);print('$v batch $b run $rep:',d['value'],d['ms_per_step'],d['roofline']['avg_launch_ms'],d['roofline']['frac'],d['parity_vs_oracle']['top100_identical'])" || tail -5 $O/bench_${v}_b${b}_$rep.err
done; done; done
cd /tmp && export TMPDIR=/tmp
for v in base keynt; do
  L=""; [ $v != base ] && L=$R/build/variants/lib_$v.so
  SIXDGS_LIB=$L timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv --kernel-include-regex "k_logits" -d $O/pmc_$v -o pmc -- python $R/bench.py --steps 1 --warmup 0 --skip-cpu-baseline --skip-reference-mode --l32-steps 0 --b8-steps 0 > $O/pmc_$v.json 2> $O/pmc_$v.err
  echo "== $v"; python $R/tools/pmc_summary.py $O/pmc_$v | tee -a $O/pmc_raw.txt
  rm -rf $O/pmc_$v
done
