# First GPU run of the experimental LDS-DMA chain kernel (k_dense_dma, dense.hip):  bash tools/dma_ab.sh   (on the GPU box, from the repo root)
#   1. bit-equality of its keys with the default kernels' (tools/cm_check.py); 2. chain throughput, alternating processes; 3. per-layer trace.
# Every step under its own timeout: a hang in a new kernel must not take the box's time limit.
O=gpurun_out/dma_ab; mkdir -p $O
timeout 90 python tools/cm_check.py save /tmp/dma_ref.pt > $O/save.log 2>&1
SIXDGS_DENSE_DMA=1 timeout 90 python tools/cm_check.py compare /tmp/dma_ref.pt > $O/compare.log 2>&1
tail -12 $O/compare.log
if grep -q "CM_CHECK PASS" $O/compare.log; then
  for i in 1 2; do
    timeout 60 python tools/time_keys.py 8388608 2>&1 | grep "planes only" | sed 's/^/default  /' >> $O/time.log
    SIXDGS_DENSE_DMA=1 timeout 60 python tools/time_keys.py 8388608 2>&1 | grep "planes only" | sed 's/^/lds-dma  /' >> $O/time.log
  done
  cat $O/time.log
  SIXDGS_DENSE_DMA=1 bash tools/trace_keys.sh dma_ab/trace > /dev/null 2>&1
  grep layer gpurun_out/dma_ab/trace/layers.txt
fi
# 4. the default kernels with the fragment reads in first-term order (private build made beforehand: python tools/build_variant.py frag -DSDG_FRAG_ORDER=1)
if [ -f build/variants/lib_frag.so ]; then
  for i in 1 2; do
    timeout 60 python tools/time_keys.py 8388608 2>&1 | grep "planes only" | sed 's/^/default    /' >> $O/frag.log
    SIXDGS_LIB=$PWD/build/variants/lib_frag.so timeout 60 python tools/time_keys.py 8388608 2>&1 | grep "planes only" | sed 's/^/frag-order /' >> $O/frag.log
  done
  cat $O/frag.log
fi
