/*
 * sixdgs.h -- C ABI of the MI355X-native 6DGS pose-estimation hot path (lib6dgs_hip.so).
 *
 * The reference (mbortolon97/6dgs) has NO FFI on this path: it is pure Python/PyTorch behind three
 * callables (generate_all_possible_rays, IdentificationModule.test_image, test_pose_estimation).
 * This header is the boundary a maintainer would bind from those callables (ctypes stub shown in
 * INTEGRATION.md); every entry point names the reference code it replaces (paths relative to the
 * reference root).
 *
 * Conventions
 *  - every pointer is a DEVICE pointer into caller-owned memory (the PyTorch caching allocator in
 *    the shipped host shim) unless the name starts with h_; the library allocates nothing
 *    persistent and keeps no mutable global state;
 *  - all tensors are fp32 row-major contiguous, indices int64, masks uint8, as in the reference;
 *  - `stream` is a hipStream_t passed as void*; every call is asynchronous on it and re-entrant;
 *  - scratch memory comes from the caller: query *_workspace_bytes, pass `ws` (256-B aligned);
 *  - return value: 0 = ok, <0 = SIXDGS_E_* argument error, >0 = hipError_t of a failed launch.
 *    No exceptions cross the ABI.  The Python shim raises RuntimeError on non-zero (the only
 *    exception type the reference driver catches, pretrain_eval_attention.py:243-244);
 *  - ragged outputs are written into caller-sized buffers and their length is returned through a
 *    device int64 (the caller syncs to read it, as the reference does at sampling.py:145).
 */
#ifndef SIXDGS_H
#define SIXDGS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SIXDGS_ABI_VERSION 7   /* 7: sixdgs_image_prep (uint8 -> resized, cropped, normalised planar fp32 in one pass); 6: sixdgs_tok_pack / sixdgs_tok_linear (dense products of the backbone stage on packed weight planes, with LayerNorm / GELU / residual fusion), sixdgs_tok_attention, sixdgs_im2col, sixdgs_u8_to_planar; the three-plane bf16 key format and its scorer kernel removed (sixdgs_split_planes, sixdgs_key_planes_bytes gone; key planes exist as scaled fp16 only); 5: sixdgs_scorer_weights carries the composite layer w4k / b4k / m4k (k_proj folded into ray-MLP layer 4 on the key-cache path), sixdgs_select_begin / _sample_stats take h_n_tok (token packing of the select sweep); 4: the select path's slack derived from |q| |k| (sixdgs_key_planes_norm_max; q + d_key_norm_max arguments) and its ray-sharded form (sample_stats / prepare / topk_u, d_uk, allow_fewer), tile maxima of U (u_tile_max); 3: sixdgs_score_select + sixdgs_select_* stages (top-k without materialised logits); 2: plane-format scorer entry points, pass1/pass2, grid kNN, split-K, distance target */
#define SIXDGS_E_BADARG (-1)
#define SIXDGS_E_WORKSPACE (-2)
#define SIXDGS_E_UNSUPPORTED (-3)

#define SIXDGS_D 384        /* embed dim (backbone.py:17, identification_module.py:44-46) */
#define SIXDGS_RAY_IN 141   /* RayPreprocessor input width (ray_preprocessor.py:15) */
#define SIXDGS_RAY_IN_PAD 144
#define SIXDGS_HID 512      /* featureC (identification_module.py:16-18) */
#define SIXDGS_TOK_IN 398   /* img_num_features + 14 (identification_module.py:20) */
#define SIXDGS_MAX_TOKENS 256

typedef void* sixdgs_stream_t;

/* How the fp32 contractions are evaluated on the matrix cores (results agree to fp32 rounding):
 *   F32     v_mfma_f32_32x32x2_f32: exact fp32 fma chain, 157 TFLOP/s peak;
 *   BF16X6  each fp32 operand split into 3 bf16 planes, the 6 leading cross terms accumulated in fp32 by
 *           v_mfma_f32_32x32x16_bf16: per-product error <= 2^-26 (below fp32 rounding), 2.67x less
 *           matrix-pipe time. */
#define SIXDGS_MMA_DEFAULT (-1)
#define SIXDGS_MMA_F32 0
#define SIXDGS_MMA_BF16X6 1
/*   F16X3   (scorer logits only) every 128-row tile of an operand scaled by a power of two and split into 2 fp16 planes,
 *           3 cross terms on v_mfma_f32_32x32x16_f16: measured error 1.0e-7 sum|a||b| (below the fp32 chain), half the
 *           MFMA work of BF16X6 and fp32-sized operands; the dense layers run BF16X6 in this mode.
 *   DEFAULT = F16X3 (scorer) + BF16X6 (dense layers). */
#define SIXDGS_MMA_F16X3 2
/*   F16X3_L32  as F16X3, but the logits travel between the two scorer passes as fp32 (1 KiB per ray and image).  F16X3
 *           stores them as 24-bit fixed point of (lane maximum - logit), resolution 2^-19 (absolute error <= 2^-20 per logit,
 *           the fp32 rounding of a logit of magnitude 16; measured effect on the scores <= 5e-7 relative): 768 B per ray and
 *           image through HBM twice, the largest data stream of the path. */
#define SIXDGS_MMA_F16X3_L32 3

/* Optional kernel timing, owned by the caller (the library stays stateless): zero-initialise, pass to
 * the *_ex entry points; each launch of the dominant kernel is bracketed by a pair of HIP events on
 * the launch stream and its algorithmic FLOP count recorded.  sixdgs_profile_collect waits for the
 * events, sums milliseconds and FLOPs, destroys the events and resets the struct. */
#define SIXDGS_PROFILE_SLOTS 128
typedef struct sixdgs_profile {
  int count;
  void* start[SIXDGS_PROFILE_SLOTS];
  void* stop[SIXDGS_PROFILE_SLOTS];
  double flops[SIXDGS_PROFILE_SLOTS];
  double bytes[SIXDGS_PROFILE_SLOTS];
} sixdgs_profile;
int sixdgs_profile_collect(sixdgs_profile* prof, double* ms_total, double* flops_total, double* bytes_total, int* launches);

int sixdgs_abi_version(void);
const char* sixdgs_error_string(int status);

/* ---------------------------------------------------------------------------------------------
 * Scene-side geometry (once per scene) -- replaces pose_estimation/sampling.py:127-267
 * ------------------------------------------------------------------------------------------- */

/* a2: mask_degraded_ellipsoids (quadricell.py:171-188) on scale = exp(log_scale)
 * (scene/gaussian_model.py:125-127).  mask[i] = total_rings(i) < target_points. */
int sixdgs_mask_degraded(const float* log_scale /*[N,3]*/, int64_t n, int target_points, uint8_t* mask /*[N]*/,
                         sixdgs_stream_t stream);

/* a5: sym_eig_3x3 (sym_eig_3x3.py:246-307), eigenvectors in the columns of vecs (may be NULL). */
int sixdgs_sym_eig_3x3(const float* mats /*[n,3,3]*/, int64_t n, float* vals /*[n,3]*/, float* vecs /*[n,3,3]*/,
                       sixdgs_stream_t stream);

/* a4: compute_normals (sampling.py:62-113): k nearest neighbours of each query in `cloud` (self
 * included), centred scatter matrix, smallest-eigenvalue eigenvector, sign by majority vote.
 * Brute force, exact; ties in distance -> lowest index.  knn (may be NULL) receives the neighbour
 * indices sorted by distance.  k <= 32. */
int sixdgs_normals_knn(const float* query /*[nq,3]*/, int64_t nq, const float* cloud /*[E,3]*/, int64_t e, int k,
                       float* normals /*[nq,3]*/, int64_t* knn /*[nq,k] or NULL*/, sixdgs_stream_t stream);
/* The same result (neighbour lists and normals bit for bit) through a uniform grid: O(E) instead of O(E^2), for
 * full-scene emission (the reference only ever runs a4 on 1000 ellipsoids).  Workspace: counting-sort buffers. */
size_t sixdgs_normals_knn_grid_workspace_bytes(int64_t e);
int sixdgs_normals_knn_grid(const float* query, int64_t nq, const float* cloud, int64_t e, int k, float* normals, int64_t* knn,
                            void* ws, size_t ws_bytes, sixdgs_stream_t stream);

/* a1+a6+a7+a10, quadricell emitter (quadricell.py:191-386 with direction_mode="isocell", SH colour
 * sampling.py:116-124,225-251).  Ellipsoid j of the emission set is Gaussian sel[j] (sel == NULL:
 * j itself); its rays are written contiguously in ellipsoid -> ring -> cell order.
 *   phase 1 (count): d_counts[j] = rays kept for ellipsoid j, d_offsets[j] = exclusive prefix,
 *                    d_total[0] = total rays, d_total[1] = total cells before the hemisphere mask;
 *   phase 2 (write): fills ori/dir/rgb/src (src = Gaussian index of each ray) using d_offsets.
 * f_dc [N,1,3] and f_rest [N,ncoef-1,3] are the raw SH tensors (gaussian_model.py:146-150).
 * `normals` [E,3] are per emission-set ellipsoid. */
/* `scale` is [N,3]: the raw log-scales of the 3DGS checkpoint when scale_is_log != 0 (the kernel
 * applies exp, gaussian_model.py:125-127), already-activated semi axes otherwise. */
int sixdgs_emit_quadricell_count(const float* xyz, const float* scale, int scale_is_log, const float* rot,
                                 const int64_t* sel, int64_t e, const float* normals, int target_points,
                                 int table_res, int64_t* d_counts /*[E]*/, int64_t* d_offsets /*[E]*/,
                                 int64_t* d_total /*[2]*/, sixdgs_stream_t stream);
int sixdgs_emit_quadricell_write(const float* xyz, const float* scale, int scale_is_log, const float* rot,
                                 const float* f_dc, const float* f_rest, int sh_degree, int n_coef,
                                 const int64_t* sel, int64_t e, const float* normals, int target_points,
                                 int table_res, const int64_t* d_offsets, float* ori, float* dir, float* rgb,
                                 int64_t* src, sixdgs_stream_t stream);
/* a6 alone (cell centres in the local frame + ellipsoid id), used by the parity tests:
 * count -> d_counts/d_offsets/d_total[0]; then centres. */
int sixdgs_quadricell_cell_counts(const float* scale /*[E,3] activated*/, int64_t e, int target_points,
                                  int64_t* d_counts, int64_t* d_offsets, int64_t* d_total, sixdgs_stream_t stream);
int sixdgs_quadricell_centers(const float* scale /*[E,3] activated*/, int64_t e, int target_points, int table_res,
                              const int64_t* d_cell_offsets /*[E]*/, float* points, int64_t* ellipsoid_id,
                              sixdgs_stream_t stream);

/* a8: isocell_distribution(ray_target, N0, isrand=-1) (isocell.py:6-84).  Returns the number of
 * directions N0*ceil(sqrt(target/N0))^2 via *h_count when dirs == NULL. */
int sixdgs_isocell_distribution(int ray_target, int n0, float* dirs /*[K,3]*/, int64_t* h_count,
                                sixdgs_stream_t stream);
/* a9: rotate_isocell (isocell.py:171-222): out[e][k] = Rodrigues(z -> normal_e) * dirs[k];
 * NaN when the normal is (anti)parallel to z, as the reference. */
int sixdgs_rotate_isocell(const float* dirs, int64_t k, const float* normals, int64_t e, float* out /*[E,K,3]*/,
                          sixdgs_stream_t stream);
/* iso-cell emitter for "every Gaussian" mode (BASELINE.json configs "64/256 isocell rays per
 * ellipsoid"; not on the reference's live path): ray (j,k): dir = rotate_isocell(dirs[k], n_j),
 * ori = centre_j + the ellipsoid surface point along dir, rgb = SH colour at -dir.  E*K rays. */
int sixdgs_emit_isocell(const float* xyz, const float* scale, int scale_is_log, const float* rot, const float* f_dc,
                        const float* f_rest, int sh_degree, int n_coef, const int64_t* sel, int64_t e,
                        const float* normals, const float* dirs, int64_t k, float* ori, float* dir, float* rgb,
                        int64_t* src, sixdgs_stream_t stream);
/* a10 alone: evaluate_viewdirs_color (sampling.py:116-124) for sh [R,3,ncoef] */
int sixdgs_eval_sh_color(const float* sh, int n_coef, const float* dirs, int64_t r, int sh_degree, float* rgb,
                         sixdgs_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Scorer, scene side (once per scene): ray MLP + k_proj -> key cache
 * replaces RayPreprocessor.forward (ray_preprocessor.py:36-46) + k_proj (our_multihead_attention.py:74)
 * ------------------------------------------------------------------------------------------- */
typedef struct sixdgs_scorer_weights {
  /* padded copies prepared once by sixdgs_pack_weights (zero padding keeps the arithmetic exact) */
  const float* w1; /* [512][144]  mlp.0  (cols 141..143 zero) */
  const float* b1; /* [512] */
  const float* w2; /* [512][512]  mlp.2 */
  const float* b2;
  const float* w3; /* [512][656]  mlp2.0 (cols 653..655 zero) */
  const float* b3;
  const float* w4; /* [384][512]  mlp2.2 */
  const float* b4;
  const float* wk; /* [384][384]  attention.k_proj */
  const float* bk;
  const float* wq; /* [384][400]  attention.q_proj (cols 398,399 zero) */
  const float* bq;
  /* max |w| of every row of w1 .. wk: the static operand scales of the scaled-fp16 x 3 dense layers (SIXDGS_MMA_DEFAULT / F16X3) */
  const float* m1; /* [512] */
  const float* m2; /* [512] */
  const float* m3; /* [512] */
  const float* m4; /* [384] */
  const float* mk; /* [384] */
  /* k_proj folded into layer 4 (no non-linearity between them: ray_preprocessor.py:27-31,46, our_multihead_attention.py:74):
   * K = (Wk W4) h3 + (Wk b4 + bk), composed in fp64 by sixdgs_pack_weights and rounded to fp32 once */
  const float* w4k; /* [384][512] */
  const float* b4k; /* [384] */
  const float* m4k; /* [384] max |w4k| per row */
  const void* planes; /* w1 .. wk, w4k pre-split into scaled fp16 planes [row][k-slabs][2][32] (w3 as [h | x padded to 160]): the operands of the
                         plane-to-plane ray MLP chain that sixdgs_ray_keys_ex runs when only keys / key planes are asked for */
} sixdgs_scorer_weights;

size_t sixdgs_packed_weights_floats(void);
/* Packs the reference state_dict tensors (SURVEY.md §8(b) shapes) into one caller buffer and fills
 * `out` with pointers into it. */
int sixdgs_pack_weights(const float* mlp0_w, const float* mlp0_b, const float* mlp2_w, const float* mlp2_b,
                        const float* mlp2_0_w, const float* mlp2_0_b, const float* mlp2_2_w, const float* mlp2_2_b,
                        const float* kproj_w, const float* kproj_b, const float* qproj_w, const float* qproj_b,
                        float* packed, sixdgs_scorer_weights* out, sixdgs_stream_t stream);

/* a12: x[R,144] = [pts, dir, rgb, PE(pts,8), PE(dir,8), PE(rgb,6), 0,0,0] */
int sixdgs_ray_encode(const float* ori, const float* dir, const float* rgb, int64_t r, float* x, sixdgs_stream_t stream);

/* a13 + k_proj.  feat (may be NULL) receives the [R,384] ray features, key the [R,384] keys.
 * Rays are processed in chunks sized by the workspace (any ws >= the minimum works). */
size_t sixdgs_ray_keys_workspace_bytes(int64_t r, int64_t max_chunk);
int sixdgs_ray_keys(const float* ori, const float* dir, const float* rgb, int64_t r, const sixdgs_scorer_weights* w,
                    float* feat, float* key, void* ws, size_t ws_bytes, sixdgs_stream_t stream);
/* same, timing the whole MLP chain of each chunk (2 025 472 algorithmic FLOP per ray) into `prof` */
/* key_planes (optional; mma_mode F16X3 / F16X3_L32 / DEFAULT only, SIXDGS_E_UNSUPPORTED otherwise): the keys as scaled fp16 planes
 * (sixdgs_key_planes_f16_bytes(r) bytes, 1536 B per ray) -- the operand of the DMA-fed scorer kernels and of the select path -- and
 * key_inv_scale[ceil(r/128)] (device, required) receives the per-tile reciprocal scales.  With key == NULL only the planes are kept.
 * (Rounds 1-5 also had a three-plane bf16 format for a bf16 x 6 plane scorer; kernel and format were removed in round 6, ABI 6.) */
/* d_key_norm_max (device scalar, may be NULL; scaled fp16 planes only): *d_key_norm_max = max(*d_key_norm_max, max over these rays of
 * |key row|), rounded up -- the bound sixdgs_score_select / sixdgs_select_candidates take; zero it before the first chunk of a scene.
 * Free when k_proj writes the planes itself (its epilogue has the rows), one pass over the planes otherwise. */
int sixdgs_ray_keys_ex(const float* ori, const float* dir, const float* rgb, int64_t r, const sixdgs_scorer_weights* w,
                       float* feat, float* key, void* key_planes, float* key_inv_scale, float* d_key_norm_max, void* ws,
                       size_t ws_bytes, sixdgs_stream_t stream, sixdgs_profile* prof, int mma_mode);
/* fp32 rows -> scaled fp16 planes [rows][12][2][32] (1536 B per row) for SIXDGS_MMA_F16X3: every 128-row tile is scaled
 * by the power of two that puts its largest magnitude in [2^13, 2^14); d_inv_scale[ceil(rows/128)] (device) receives the
 * reciprocal scales.  Splitting a row range in chunks is valid when every chunk starts at a multiple of 128 rows. */
size_t sixdgs_key_planes_f16_bytes(int64_t r);
int sixdgs_split_planes_f16(const float* src, int64_t rows, int64_t ld, void* planes, float* d_inv_scale, sixdgs_stream_t stream);

/* generic fp32 MFMA GEMM used by the above: y[M,N] = act(x[M,K] . w[N,K]^T + b), K % 16 == 0,
 * N % 128 == 0, ldx/ldw/ldy in floats and multiples of 4. */
int sixdgs_linear(const float* x, int64_t m, int k, int64_t ldx, const float* w, int64_t ldw, const float* b, int n,
                  int relu, float* y, int64_t ldy, sixdgs_stream_t stream);
int sixdgs_linear_ex(const float* x, int64_t m, int k, int64_t ldx, const float* w, int64_t ldw, const float* b, int n,
                     int relu, float* y, int64_t ldy, sixdgs_stream_t stream, int mma_mode);
/* The same product with K cut into `slices` parts computed by separate workgroups and added in ascending order
 * (deterministic): for few output tiles and a long K (the camera-up CNN as im2col GEMMs: M <= a few hundred, K = 9600). */
size_t sixdgs_linear_splitk_workspace_bytes(int64_t m, int n, int slices);
int sixdgs_linear_splitk(const float* x, int64_t m, int k, int64_t ldx, const float* w, int64_t ldw, const float* b, int n,
                         int relu, float* y, int64_t ldy, int slices, void* ws, size_t ws_bytes, sixdgs_stream_t stream,
                         int mma_mode);

/* The dense products of the backbone stage (pose_estimation/backbone.py:82-114 runs DINOv2 ViT-S/14 on every query image) with the
 * elementwise work around them folded in: y = epilogue( prologue(x) . w^T + bias ), fp32 results (two scaled fp16 planes per operand, three cross terms,
 * fp32 accumulation), tiles of 64 token rows x 256 features -- sized for token matrices of 257 .. 4112 rows.  The weights are constants and are split
 * ONCE: sixdgs_tok_pack turns w [n][ldw] (n a multiple of 128, k a multiple of 384) into sixdgs_tok_pack_bytes(n, k) bytes of planes in the matrix
 * pipe's operand order + n reciprocal row scales; sixdgs_tok_linear takes those.
 *   a_mode   SIXDGS_TOK_A_PLAIN  x [m][ldx];
 *            SIXDGS_TOK_A_LAYERNORM  LayerNorm(x; ln_weight, ln_bias, ln_eps) over the k = 384 columns in front of the product (torch.nn.LayerNorm,
 *                                dinov2 block.norm1 / norm2);
 *   epilogue SIXDGS_TOK_EPI_BIAS  y [m][ldy] = acc + bias;   SIXDGS_TOK_EPI_GELU  gelu(acc + bias), erf form (mlp.fc1 + act; erf to 1.5e-7);
 *            SIXDGS_TOK_EPI_RESID  residual [m][ldr] + gamma[n] * (acc + bias) (x + ls(branch(x)): attn.proj / mlp.fc2, LayerScale gamma or NULL = 1;
 *                                y may be the residual buffer itself). */
#define SIXDGS_TOK_A_PLAIN 0
#define SIXDGS_TOK_A_LAYERNORM 1
#define SIXDGS_TOK_EPI_BIAS 0
#define SIXDGS_TOK_EPI_GELU 1
#define SIXDGS_TOK_EPI_RESID 2
size_t sixdgs_tok_pack_bytes(int n, int k);
int sixdgs_tok_pack(const float* w /*[n][ldw]*/, int n, int k, int64_t ldw, void* planes, float* inv_scale /*[n]*/, sixdgs_stream_t stream);
int sixdgs_tok_linear(const float* x, int64_t m, int k, int64_t ldx, int a_mode, const float* ln_weight, const float* ln_bias, float ln_eps,
                      const void* w_planes, const float* w_inv_scale, const float* bias /*[n] or NULL*/, int n, int epilogue, const float* residual,
                      int64_t ldr, const float* gamma, float* y, int64_t ldy, sixdgs_stream_t stream);

/* The attention of a ViT block (dinov2 Attention.forward: softmax(q k^T / sqrt(64)) v per image and head) on the QKV product's output as it lies:
 * qkv [images * tokens][ldq] = q | k | v, each heads * 64 wide, head h at columns h * 64; y [images * tokens][ldy], head h at columns h * 64 (what
 * attn.proj reads).  Head dimension 64, tokens <= 288 (SIXDGS_E_UNSUPPORTED beyond: the caller keeps its own attention); fp32-class results. */
int sixdgs_tok_attention(const float* qkv, int64_t ldq, int images, int tokens, int heads, float* y, int64_t ldy, sixdgs_stream_t stream);

/* a22 (camera_direction_network.py:29-36, the valid k x k convolutions of the camera-up CNN) as GEMMs: the im2col matrix of a whole batch in one
 * launch.  a [batch * ho * wo][channels * k * k] (ho = height - k + 1, wo = width - k + 1): row (b, oy, ox); column (c, ky, kx) -- the order of
 * conv.weight.view(out, -1) -- or, taps_major != 0, (ky, kx, c) (for weights whose columns the caller permuted the same way: with channel-contiguous
 * input every (row, tap) is then a copy of `channels` consecutive floats); value x[b * stride_b + c * stride_c + (oy + ky) * stride_y + (ox + kx) * stride_x],
 * strides in elements, free (NCHW, or the [B * ho * wo][C] output of the previous layer's GEMM read in place). */
int sixdgs_im2col(const float* x, int64_t stride_b, int64_t stride_c, int64_t stride_y, int64_t stride_x, int batch, int channels, int height, int width, int k,
                  int taps_major, float* a, sixdgs_stream_t stream);

/* a16, first step, for a batch (pose_estimation/test.py:69-73: uint8 image / 255.0): images [batch][pixels][3] uint8 -> out [batch][3][pixels] fp32,
 * out = table256[value] (the caller's table carries the reference's true division); pixels a multiple of 4. */
int sixdgs_u8_to_planar(const uint8_t* images, int batch, int64_t pixels, const float* table256, float* out, sixdgs_stream_t stream);

/* a16 + the wrapper's transform pipeline for a batch of RGB images of one size (pose_estimation/test.py:69-73, backbone.py:52-77: uint8 / 255.0 -> Resize(256,
 * bicubic, antialias) -> CenterCrop(224) -> Normalize) in one pass: images [batch][height][width][3] uint8 -> out [batch][3][out_size][out_size] fp32 =
 * (crop(resize_aa(table256[value])) - mean3[c]) / std3[c].  The caller states the geometry the reference's transforms would use: the resized grid
 * (resized_h, resized_w) and the crop's corner in it.  The arithmetic is the antialiased bicubic of the op it replaces (spans, a = -0.5 filter, tap-order sums, rows
 * reduced along x first), with the x-reductions shared between the output rows of a band; mean3 / std3 are HOST arrays (read at the call).  SIXDGS_E_UNSUPPORTED for
 * scales beyond 15 or windows that do not fit the kernel's LDS: the caller keeps PyTorch's kernels for those. */
int sixdgs_image_prep(const uint8_t* images, int batch, int height, int width, const float* table256, int resized_h, int resized_w, int crop_top, int crop_left,
                      int out_size, const float* mean3, const float* std3, float* out, sixdgs_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Scorer, image side (per batch of query images)
 * replaces MultiHeadAttention.forward (our_multihead_attention.py:70-79, 4-12),
 * IdentificationModule.run_attention's column sum (identification_module.py:80-82) and
 * torch.topk (identification_module.py:131)
 * ------------------------------------------------------------------------------------------- */
/* q[b] = tokens[b] . Wq^T + bq.  tokens [B, 256, 398] (rows >= n_tok[b] ignored), q [B,256,384]
 * (rows >= n_tok[b] are written as zeros).  d_n_tok: device int32 [B], 0 <= n_tok <= 256. */
int sixdgs_q_proj(const float* tokens, const int32_t* d_n_tok, int batch,
                  const sixdgs_scorer_weights* w, float* q, sixdgs_stream_t stream);

/* scores[b][r] = sum_t softmax_r(q[b][t] . key[r] / sqrt(384)); idx/val = top-k (sorted
 * descending, ties -> lowest index).  scores may be NULL (then they live only in the workspace).
 * Never materialises more than `ws` allows: images are processed in groups that fit. */
size_t sixdgs_score_topk_workspace_bytes(int64_t r, int batch, int topk);   /* enough for every mode */
/* exact for a mode: with key planes in F16X3 / DEFAULT the logits take 784 instead of 1024 B per ray and image */
size_t sixdgs_score_topk_workspace_bytes_ex(int64_t r, int batch, int topk, int mma_mode, int with_key_planes);
int sixdgs_score_topk(const float* q /*[B,256,384]*/, const int32_t* d_n_tok, int batch, const float* key /*[R,384]*/,
                      int64_t r, int topk, float* scores /*[B,R] or NULL*/, int64_t* idx /*[B,topk]*/,
                      float* val /*[B,topk]*/, float* row_stats /*[B,256,2] (max, sumexp) or NULL*/, void* ws,
                      size_t ws_bytes, sixdgs_stream_t stream);
/* same, timing each launch of the logits kernel (2*T*384 algorithmic FLOP per ray and image) into `prof` */
/* key_planes != NULL in the F16X3 / F16X3_L32 / DEFAULT modes selects the DMA-fed kernel: scaled fp16 planes + d_key_scale (the d_inv_scale of
 * sixdgs_split_planes_f16 / sixdgs_ray_keys_ex); `key` (fp32) may then be NULL.  SIXDGS_MMA_F32 / _BF16X6 score on `key` (fp32 MFMA chain / bf16 x 6
 * with the split on the fly) and ignore the planes; planes without `key` in those modes: SIXDGS_E_UNSUPPORTED. */
int sixdgs_score_topk_ex(const float* q, const int32_t* d_n_tok, const int32_t* h_n_tok /*host copy, for the FLOP count*/,
                         int batch, const float* key, const void* key_planes, const float* d_key_scale, int64_t r, int topk,
                         float* scores, int64_t* idx, float* val, float* row_stats, void* ws, size_t ws_bytes,
                         sixdgs_stream_t stream, sixdgs_profile* prof, int mma_mode);

/* Ray-sharded scoring (SURVEY 8(e) fallback: the key cache of a scene is split across GPUs, or one image must be scored by
 * several).  The softmax runs over ALL rays, so the scorer is cut at the row statistics:
 *   pass 1  logits of this shard's rays for ALL `batch` images stay in the workspace (SIXDGS_E_WORKSPACE if they do not
 *           fit: sixdgs_score_topk_workspace_bytes(r, batch, topk)); row_stats[B,256,2] receives the shard's (max, sumexp);
 *   caller  combines the shards: M = max over shards, S = sum over shards of s * exp(m - M)  (two tiny all-reduces);
 *   pass 2  takes the global statistics, writes this shard's scores and its local top-k (indices local to the shard);
 *           the global top-k is the (value desc, global index asc) merge of the shards' candidates.
 * Same r, batch, topk, workspace and mma_mode in both passes; used_planes = pass 1 ran on key planes. */
int sixdgs_score_pass1(const float* q, const int32_t* d_n_tok, const int32_t* h_n_tok, int batch, const float* key,
                       const void* key_planes, const float* d_key_scale, int64_t r, int topk, float* row_stats, void* ws,
                       size_t ws_bytes, sixdgs_stream_t stream, sixdgs_profile* prof, int mma_mode);
int sixdgs_score_pass2(const float* row_stats, const int32_t* d_n_tok, int batch, int used_planes, int64_t r, int topk,
                       float* scores, int64_t* idx, float* val, void* ws, size_t ws_bytes, sixdgs_stream_t stream, int mma_mode);
/* Top-k WITHOUT materialising the logits -- the inference path, where only idx/val are wanted (the reference driver reads nothing
 * else: test.py:105-107).  Needs the scaled fp16 key planes of ALL r rays (sixdgs_ray_keys_ex, SIXDGS_MMA_F16X3) and those of a
 * ray SAMPLE (any r_sample <= r rays of the same scene, e.g. one ray in 16, through the same entry point).  score[r] =
 * sum_t e[t][r] / Z_t with e = exp(logit - ref_t), Z_t = sum_r e[t][r]:
 *   pre-pass    over the sample: ref_t = sample maximum, Z~_t = sample sum (so that f Z~_t ~ Z_t, f = r / r_sample).  Since round 6 with ONE of the
 *               three MFMA terms (h x h: logits to ~2^-11 |q||k| / sqrt(384); SIXDGS_PREPASS_TERMS=3 restores all three): these two only set the sweep's
 *               exponent offsets and scale -- g_t below is exact relative to WHATEVER Z~_t the sweep was given;
 *   main sweep  over all rays, one matrix-core pass, nothing of size T x R leaves the chip: U[r] = sum_t e[t][r] / (f Z~_t)
 *               (4 x 4 B per ray and image instead of 784 B of logits) and the EXACT g_t = Z_t / (f Z~_t);
 *   bounds      score[r] = sum_t e'[t][r] / g_t lies in [U[r] / g_max, U[r] / g_min], so every ray of the true top-k has
 *               U[r] >= U_(k) g_min / g_max (1 - eps) / (1 + eps) (U_(k) = k-th largest U).  eps bounds the relative difference between
 *               the sweep's U and the exact re-score: eps = 1.4e-4 x + 1.3e-5 with x = max_t |q_t| max_r |k_r| / sqrt(384) >= every
 *               sum_i |q_i k_i| / sqrt(384) (fp32 accumulation of 1152 products in the MFMAs + the epilogue's roundings; derivation at
 *               k_sel_bounds).  d_key_norm_max (device scalar): max_r |k_r| of the scene, from sixdgs_key_planes_norm_max;
 *   re-score    those candidates exactly (fp32, from their key planes and the exact g_t); their top-k (value descending, ties ->
 *               lowest index) is the result.  The sample decides only how many candidates there are, never the answer.
 * d_status[b] (device) = number of candidates examined, or -1 when this image must be scored by sixdgs_score_topk_ex instead
 * (more than max_candidates candidates, or an exponent overflow because a logit exceeds the sample maximum by > 88).
 * max_candidates: multiple of 8, >= topk.  Workspace ~ 20 B per ray and image. */
size_t sixdgs_score_select_workspace_bytes(int64_t r, int batch, int topk, int max_candidates);
/* *d_norm_max = max(*d_norm_max, max over the rows of |x_row|) for the values the scaled fp16 planes hold (rounded up: it is used as a
 * bound).  Zero *d_norm_max before the first call; scenes that go through in chunks accumulate chunk by chunk. */
int sixdgs_key_planes_norm_max(const void* planes, const float* d_scale, int64_t rows, float* d_norm_max, sixdgs_stream_t stream);
/* The four stages of the select path as entry points of their own, for scenes whose key planes do not fit the GPU: `begin` once
 * (sample pre-pass), `sweep` per ray chunk (chunks start at multiples of 256 rays; each writes its columns of U and adds its
 * share of the exact per-token sums into gsum), `candidates` once over the whole U, then `rescore` on the key planes of the
 * candidates alone.  All buffers are the caller's: ctok, gsum [B,256] floats; U [B][u_stride] floats (u_stride >= r rounded up
 * to 256); cand [B][max_candidates] int64 (ascending ray indices); d_count [B] int32 (candidates, may exceed max_candidates;
 * -1 bounds unusable; -2 image without tokens).  sixdgs_select_workspace_bytes(r, ...) with the largest r of any call.
 * rescore: `planes` are either the scene's key planes (compact == 0: rows addressed by ray index, scales per 128 rays) or the
 * planes of exactly the candidates in candidate order (compact != 0: row b * max_candidates + c, scales per 128 ROWS), e.g.
 * from sixdgs_ray_keys_ex on the gathered rays. */
size_t sixdgs_select_workspace_bytes(int64_t r, int batch, int topk, int max_candidates);              /* begin, sweep: r = rays of the call */
size_t sixdgs_select_candidates_workspace_bytes(int64_t r, int batch, int topk, int max_candidates);   /* candidates (r = all rays), rescore */
/* h_n_tok (begin, sample_stats, sweep, sixdgs_score_select; may be NULL): a HOST copy of d_n_tok.  With it the images of a launch are PACKED by their
 * token counts into the 256-token tiles of the matrix-core sweep (two views of <= 128 tokens, four of <= 64, 192 + 64 ... share a tile; ABI 5), so a
 * masked view costs what its surviving tokens cost (the reference scores only those: backbone.py:86-114, identification_module.py:80-82).  It MUST equal
 * d_n_tok; an image the host copy gives too few tokens is reported undecidable (status -1), never scored without some of its tokens.  Results do not
 * depend on it: an image's U, sums and statistics are the same bits packed or alone.  NULL: one image per tile (and 256 tokens each in the FLOP count). */
/* How sixdgs_select_sweep / _sample_stats / sixdgs_score_select cut `batch` images with these token counts (NULL: unknown) into launches: returns the
 * number of launches (>= 0; < 0 an error) and fills, for the first max_launches of them, the 256-token tiles ("slots") and the images of each.  Host
 * arithmetic only (csrc/sweep_plan.h); for reporting and tests. */
int sixdgs_select_sweep_plan(const int32_t* h_n_tok, int batch, int32_t* slots_per_launch, int32_t* images_per_launch, int max_launches);
int sixdgs_select_begin(const float* q, const int32_t* d_n_tok, const int32_t* h_n_tok, int batch, const void* sample_planes, const float* d_sample_scale,
                        int64_t r_sample, int64_t r_total, float* ctok, float* gsum, void* ws, size_t ws_bytes, sixdgs_stream_t stream);
/* Ray-sharded select (the scene's key planes split over ranks, SURVEY 8(e) fallback): `begin` in two halves, so that the shards can
 * merge their sample statistics in between -- sample_stats writes this shard's (max, sumexp) [B,256,2] of ITS sample; the caller
 * merges (M = max, S = sum s e^(m - M): two all-reduces of 1 KB per image) and hands the global statistics to prepare together with
 * the TOTAL sample and ray counts.  Then per shard: sweep -> all-reduce(SUM) of gsum, all-reduce(MAX) of the key norm ->
 * sixdgs_select_topk_u (the shard's k largest U, descending, NaN-padded) -> all-gather, k-th largest of the union = U_(k) of the
 * scene -> candidates with d_uk [B] -> rescore with allow_fewer (a shard may hold fewer than k candidates: idx / val are then padded
 * with -1 / NaN and status = candidates examined) -> all-gather + (value desc, global index asc) merge. */
int sixdgs_select_sample_stats(const float* q, const int32_t* d_n_tok, const int32_t* h_n_tok, int batch, const void* sample_planes, const float* d_sample_scale,
                               int64_t r_sample, float* row_stats /*[B,256,2]*/, void* ws, size_t ws_bytes, sixdgs_stream_t stream);
int sixdgs_select_prepare(const float* row_stats, const int32_t* d_n_tok, int batch, int64_t r_sample, int64_t r_total, float* ctok, float* gsum,
                          sixdgs_stream_t stream);
int sixdgs_select_topk_u(const float* u, int64_t u_stride, int64_t r, const float* u_tile_max /*or NULL*/, int batch, int topk, float* val /*[B,topk]*/,
                         void* ws, size_t ws_bytes, sixdgs_stream_t stream);        /* ws: sixdgs_select_candidates_workspace_bytes */
/* u_tile_max (optional, [B][u_stride / 256] floats, u_stride a multiple of 256; a chunk passes u + ray_offset and u_tile_max + ray_offset / 256):
 * the largest U of every 256-ray tile.  The k-th largest tile maximum is a lower bound of the k-th largest U -- k tiles hold a ray that large --
 * and with the top rays scattered over r / 256 tiles practically equal to it; `candidates` (and `topk_u`) take it from these r / 256 values
 * instead of a radix select over all r values of U (six passes less over U per batch).  A lower threshold admits a few more candidates, never fewer. */
int sixdgs_select_sweep(const float* q, const int32_t* d_n_tok, const int32_t* h_n_tok, int batch, const void* key_planes,
                        const float* d_key_scale, int64_t r, const float* ctok, float* gsum, float* u, int64_t u_stride, float* u_tile_max,
                        void* ws, size_t ws_bytes, sixdgs_stream_t stream, sixdgs_profile* prof);
int sixdgs_select_candidates(const float* u, int64_t u_stride, int64_t r, const float* u_tile_max /*or NULL*/, const float* q, const int32_t* d_n_tok, int batch, const float* gsum,
                             const float* d_key_norm_max, const float* d_uk /*[B] k-th largest U of the whole scene, or NULL = of these rays*/,
                             int topk, int max_candidates, int64_t* cand, int32_t* d_count, void* ws, size_t ws_bytes, sixdgs_stream_t stream);
int sixdgs_select_rescore(const float* q, const int32_t* d_n_tok, int batch, const void* planes, const float* d_scale, int compact,
                          const float* ctok, const float* gsum, const int64_t* cand, const int32_t* d_count, int64_t r, int topk,
                          int max_candidates, int allow_fewer, int64_t* idx, float* val, int32_t* d_status, void* ws, size_t ws_bytes,
                          sixdgs_stream_t stream);
int sixdgs_score_select(const float* q, const int32_t* d_n_tok, const int32_t* h_n_tok /*host copy for the FLOP count, may be NULL*/,
                        int batch, const void* key_planes, const float* d_key_scale, const float* d_key_norm_max, int64_t r,
                        const void* sample_planes, const float* d_sample_scale, int64_t r_sample, int topk, int max_candidates, int64_t* idx /*[B,topk]*/,
                        float* val /*[B,topk]*/, int32_t* d_status /*[B]*/, void* ws, size_t ws_bytes, sixdgs_stream_t stream,
                        sixdgs_profile* prof);
/* top-k alone over precomputed scores [B,R] */
size_t sixdgs_topk_workspace_bytes(int64_t r, int batch, int topk);
int sixdgs_topk(const float* scores, int64_t r, int batch, int topk, int64_t* idx, float* val, void* ws,
                size_t ws_bytes, sixdgs_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Pose assembly (per image) -- replaces pose_estimation/test.py:157-198,216-218 with
 * line_intersection.py:5-34,75-154 and error_computation.py:3-8
 * ------------------------------------------------------------------------------------------- */
/* SURVEY 8(f)#1, forward part: the target scores of DistanceBasedScoreLoss (distance_based_loss.py:5-71,179-222) for the
 * ground-truth camera d_pose (device, row-major c2w 4x4): 1 - tanh(distance of the camera centre to each ray), zero for ray
 * origins behind the camera plane, rescaled so that the targets sum to n_tokens.  d_sum (device scalar, may be NULL)
 * receives the un-scaled sum.  The loss value mean((pred - target)^2) and its gradient stay with the caller (PyTorch). */
size_t sixdgs_distance_target_workspace_bytes(int64_t r);
int sixdgs_distance_target(const float* rays_ori, const float* rays_dir, int64_t r, const float* d_pose, int n_tokens,
                           float* target /*[r]*/, float* d_sum, void* ws, size_t ws_bytes, sixdgs_stream_t stream);

/* For each image b: duplicate-origin filter, unweighted LS centre (NaN when det < 1e-7),
 * exclude_negatives reweighting, watch direction, make_rotation_mat(-watch, up[b]), singular -> I,
 * c2w = [inv(R) | centre], NaN -> I4.
 * outputs: c2w [B,4,4]; status [B] bit0 = singular rotation, bit1 = NaN pose (identity returned),
 * bit2 = NaN centre; w_final [B,k] (0 for filtered rays), n_kept [B]; errors [B,2] =
 * (translation error, angular error in degrees) against gt_c2w when gt_c2w != NULL. */
int sixdgs_solve_pose(const float* rays_ori, const float* rays_dir, int64_t r, const int64_t* idx /*[B,k]*/,
                      const float* val /*[B,k]*/, int k, const float* up /*[B,3]*/, const float* gt_c2w /*[B,4,4]*/,
                      int batch, float* c2w, int32_t* status, float* w_final, int32_t* n_kept, float* centre /*[B,3]*/,
                      float* errors, sixdgs_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SIXDGS_H */
