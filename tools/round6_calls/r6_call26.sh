# Round 6, call 26: cycle stamps of the chain's layers (slab loop vs epilogue of one persistent workgroup), profiling build.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06c26; mkdir -p $O
SIXDGS_LIB=$GRAFT_REPO_ROOT/build/variants/lib_denseprof.so python -W ignore tools/prof_dense.py 2>&1 | grep -v amdgpu.ids | tee $O/prof_dense.log
