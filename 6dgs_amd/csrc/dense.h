// dense.h -- internal interface between gemm.hip (weight packing, sixdgs_ray_keys_ex) and dense.hip (the plane-to-plane ray MLP chain)
#pragma once
#include "common.h"

namespace sdg {
// bytes of the pre-split weight planes of the five ray-side layers (appended to the packed weight buffer)
size_t dense_weight_plane_bytes();
// fp32 weights + row maxima of `w` (already packed) -> scaled fp16 planes
int dense_pack_weight_planes(const sixdgs_scorer_weights* w, char* planes, hipStream_t s);
// scratch bytes per ray of dense_chain
size_t dense_chain_bytes_per_ray();
// rays -> through the plane-to-plane layers -> fp32 keys kdst [m][384], or (kplanes != null; m's start a multiple of 128 rays) the scorer's
// fp16 key planes [m][1536 B] with kinv[tile of 128 rays] = reciprocal tile scale, identical to sixdgs_split_planes_f16 of the fp32 keys
// knorm_max (with kplanes; may be null): *knorm_max = max(*knorm_max, max over the m rays of |key row|), rounded up -- from the k_proj epilogue
int dense_chain(const float* ori, const float* dir, const float* rgb, int64_t m, const sixdgs_scorer_weights* w, const char* wplanes, float* kdst, char* kplanes,
                float* kinv, float* knorm_max, char* ws, hipStream_t s);
}  // namespace sdg
