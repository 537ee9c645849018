# Round 6, call 18: the image side as replayed, fused off / on.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06c18; mkdir -p $O
cd $R
for v in 0 1; do echo "SIXDGS_VIT_FUSED=$v"; SIXDGS_VIT_FUSED=$v python -W ignore tools/time_image_side.py 2>&1 | grep -v amdgpu.ids | tee $O/image_side_fused$v.md; done
