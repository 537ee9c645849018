# Round 6, call 29: k_solve_pose on four waves: sections, parity tests.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06c29; mkdir -p $O
export SIXDGS_RANDOM_BACKBONE=1
cd $R
SIXDGS_LIB=$R/build/variants/lib_poseprof.so python -W ignore tools/prof_pose.py 2>&1 | grep -v amdgpu.ids | tee $O/prof_pose.log
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_e2e.py tests/test_gpu_cfg1.py -q -x 2>&1 | tail -n 8 ) > $O/tests.log 2>&1
grep -E "passed|failed|Error|assert " $O/tests.log | head
