# Round 5, call 12: would a tile-major key-plane layout ([256-ray tile][slab][ray][128 B]: a slab = 32 KB of consecutive bytes) feed the sweep faster?
# A timing-only build reads the planes in that pattern (same bytes, garbage results) against the product build, alternating, same box.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05c12; mkdir -p $O
for i in 1 2; do
  for L in "" "$GRAFT_REPO_ROOT/build/variants/lib_ktm.so"; do
    for BR in "1 19200000" "2 32000000" "4 32000000" "8 32000000"; do
      SIXDGS_LIB=$L python tools/time_sweep.py $BR 3 2>&1 | tail -1 | sed "s|^|lib=${L##*/} |" | tee -a $O/key_tile_major_ab.log
    done
  done
done
