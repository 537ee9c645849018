# FETCH_SIZE and HIP-event time of the select sweep per grid layout, repeated: bash tools/sib_ab2.sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03q2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
for cfg in "0 1" "2 1" "1 1" "1 4"; do
  set -- $cfg; export SIXDGS_SIBLING_SYNC=$1 SIXDGS_SIB_PERIOD=$2
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv --kernel-include-regex "k_logits" -d $O/pmc -o pmc -- python $R/bench.py --steps 1 --warmup 0 --skip-cpu-baseline --skip-reference-mode --l32-steps 0 > $O/b.json 2>/dev/null
  echo "mode=$1 period=$2 rep=$rep $(python $R/tools/pmc_summary.py $O/pmc | grep '0, 3' | awk '{print $NF}') launch_ms=$(python -c "import json;print(json.load(open('$O/b.json'))['roofline']['avg_launch_ms'])")"
  rm -rf $O/pmc
done
done
