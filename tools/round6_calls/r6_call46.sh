# Round 6, call 46: `nt` key loads as the default -- select tests, default against -DSDG_KEY_AUX=0 on the headline (8 and 4 images), cfg-2, cfg-3; FETCH_SIZE / WRITE_SIZE of the default.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06c46; mkdir -p $O
export SIXDGS_RANDOM_BACKBONE=1
cd $R
( timeout 1200 python -m pytest tests/test_gpu_select.py tests/test_gpu_parity.py -q -x 2>&1 | tail -n 4 ) > $O/tests.log 2>&1
grep -E "passed|failed|Error|assert " $O/tests.log | head
for rep in 1 2; do for v in nt plain; do for c in "headline --batch 8" "headline --batch 4" cfg2 cfg3; do
  L=""; [ $v = plain ] && L=$R/build/variants/lib_keyplain.so
  n=$(echo $c | tr -d ' -'); st=10; [ "$c" = cfg2 ] && st=40; [ "$c" = cfg3 ] && st=5
  SIXDGS_LIB=$L python -W ignore bench.py --config $c --steps $st --warmup 2 --skip-cpu-baseline --skip-reference-mode --l32-steps 0 --b8-steps 0 > $O/bench_${v}_${n}_$rep.json 2> $O/bench_${v}_${n}_$rep.err
  python -c "
import json;d=json.loads([l for l in open('$O/bench_${v}_${n}_$rep.json') if l.startswith('{')][-1]);print('$v $n run $rep:',d['value'],d['ms_per_step'],d['median_step']['ms'],d['roofline']['avg_launch_ms'],d['roofline']['frac'])" || tail -5 $O/bench_${v}_${n}_$rep.err
done; done; done
cd /tmp && export TMPDIR=/tmp
for c in "headline --batch 8" "headline --batch 4" cfg3; do for C in FETCH_SIZE WRITE_SIZE; do
  n=$(echo $c | tr -d ' -')
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv --kernel-include-regex "k_logits|k_sel_finish" -d $O/pmc_$n -o pmc -- python $R/bench.py --config $c --steps 1 --warmup 0 --skip-cpu-baseline --skip-reference-mode --l32-steps 0 --b8-steps 0 > $O/pmc_$n.json 2> $O/pmc_$n.err
  echo "== $n" | tee -a $O/pmc_raw.txt; python $R/tools/pmc_summary.py $O/pmc_$n | tee -a $O/pmc_raw.txt
  rm -rf $O/pmc_$n
done; done
