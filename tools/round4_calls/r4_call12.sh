# Round 4: the whole-step hipGraph at one image per step replays in 27 ms plainly and in 13.6 ms under rocprofv3 --hip-trace.  Which runtime setting decides?
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4c12; mkdir -p $O
run() { n=$1; shift; (env "$@" timeout 200 python bench.py --config cfg2 --graph --steps 12 --warmup 3 --skip-cpu-baseline --skip-reference-mode --l32-steps 0 > $O/g_$n.json 2> $O/g_$n.err); python -c "
import json;d=json.load(open('$O/g_$n.json'));print('$n', d['value'], d['ms_per_step'], d['median_step']['ms'], d['median_step']['min_ms'], d['median_step']['max_ms'])" 2>&1 | tail -1; }
run plain A=1
run packet_capture_1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run packet_capture_0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run one_hw_queue GPU_MAX_HW_QUEUES=1
run active_wait HIP_FORCE_DEV_KERNARG=1 HSA_ENABLE_INTERRUPT=0
