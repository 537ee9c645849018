// sweep_plan.h -- how a batch of images is cut into launches of the select sweep (host-side arithmetic; shared with the CPU test-suite through hostcheck.cpp).
#pragma once

namespace sdg {

// Images of the NEXT launch when `left` images of a batch remain: launches of `cap`, the last one taking what is left, up to cap + cap / 2 (a launch of one
// or two images would pull every key tile from HBM for itself).  cap <= 0: the whole batch in one launch.  Why a cap at all: sixdgs_select_sweep, score.hip.
inline int sweep_launch_images(int left, int cap) {
  if (cap <= 0) return left;
  const int tail = cap + cap / 2;
  return left > tail ? cap : left;
}

}  // namespace sdg
