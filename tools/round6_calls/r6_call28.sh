# Round 6, call 28: where k_solve_pose's 150 us go.
cd $GRAFT_REPO_ROOT; SIXDGS_LIB=$GRAFT_REPO_ROOT/build/variants/lib_poseprof.so python -W ignore tools/prof_pose.py 2>&1 | grep -v amdgpu.ids
