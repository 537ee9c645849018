# Round 4, first GPU call: (1) the LDS-DMA chain kernel's first run (bit-equality, A/B), (2) power / clock traces, (3) RCCL at world size 1.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4c1
bash tools/dma_ab.sh > gpurun_out/r4c1/dma_ab.log 2>&1; tail -30 gpurun_out/r4c1/dma_ab.log
for w in idle:4 sweep:10 chain:10 copy:6; do
  timeout 120 python tools/power_trace.py ${w%%:*} ${w##*:} 2>&1 | grep -E "POWER|Error|error" | tail -3
done
SIXDGS_LIB=$PWD/build/variants/lib_abl.so timeout 120 python tools/power_trace.py mfma 10 2>&1 | grep -E "POWER|Error|error" | tail -3
amd-smi static -l 2>&1 | head -30 > gpurun_out/r4c1/amdsmi_static_limit.txt
amd-smi metric -p -c 2>&1 | head -60 > gpurun_out/r4c1/amdsmi_metric.txt
rocm-smi --showpower --showclocks --showmaxpower 2>&1 | head -40 > gpurun_out/r4c1/rocmsmi.txt
for extra in "" "--select"; do
  SIXDGS_DIST_BACKEND=nccl timeout 180 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 tools/ray_shard_check.py --backend nccl $extra > gpurun_out/r4c1/nccl_ws1$extra.log 2>&1
  echo "nccl ws1 $extra rc=$?"; tail -4 gpurun_out/r4c1/nccl_ws1$extra.log
done
