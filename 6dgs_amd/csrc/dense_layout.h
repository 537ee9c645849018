// dense_layout.h -- index arithmetic of the ray-MLP chain's operand planes (dense.hip), usable from device AND host code: the CPU test-suite
// walks a tag through store -> HBM -> load -> LDS -> MFMA fragment with these very functions (libsixdgs_hostcheck.so,
// tests/test_dense_layout.py), so the layouts of producer and consumer cannot drift apart unnoticed on a machine without a GPU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sdg {
namespace dl {

#define SDG_DL __host__ __device__ __forceinline__ constexpr

constexpr int kSlabB = 128;              // bytes of one (row, slab): plane h 64 B, plane l 64 B; a slab = 32 inputs (k)
constexpr int kPRow = 144;               // LDS row stride of a staged slab (128 B + 16: consecutive rows start 36 banks apart)
constexpr int kGran = 128;               // rays per granule of chunk-major planes
constexpr int kGranSlab = kGran * 128;   // bytes of one (granule, slab)
constexpr int kChunkRun = kGran * 16;    // bytes of one (granule, slab, plane, chunk): 128 rays x 16 B

// Chunk-major layers permute the 32 rows of every MFMA row block: MFMA row m = 8 rg + 4 h + j (register group rg, lane half h, register j)
// computes feature pi(m) = 16 (rg >> 1) + 8 h + 4 (rg & 1) + j of the block (bits 2 and 3 of m swapped; an involution), so that lane half h
// holds the features 16 p + 8 h + 0..7 -- chunk 2 p + h of the 32-feature slab -- in its register groups 2 p and 2 p + 1.
SDG_DL int row_perm(int m) { return (m & ~12) | ((m & 4) << 1) | ((m & 8) >> 1); }

// MFMA 32x32x16 f16 result: accumulator register r (0..15) of lane l is row 8 (r >> 2) + 4 (l >> 5) + (r & 3), column l & 31
SDG_DL int acc_row(int lane, int r) { return 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3); }
// feature (within the wave's 32-feature slab) of that register: the row itself, or -- permuted weight rows -- row_perm of it
SDG_DL int acc_feature(bool permuted, int lane, int r) { return permuted ? row_perm(acc_row(lane, r)) : acc_row(lane, r); }

// byte offset of the 16-byte chunk (plane, c) -- inputs 8 c .. 8 c + 7 of slab `slab` -- of ray `ray` in chunk-major planes of nslab slabs per ray
SDG_DL int64_t cm_offset(int64_t ray, int nslab, int slab, int plane, int c) {
  return ((ray >> 7) * nslab + slab) * (int64_t)kGranSlab + (plane * 4 + c) * kChunkRun + (ray & 127) * 16;
}
// the same in ray-major planes [ray][slab][plane h 64 B | plane l 64 B]
SDG_DL int64_t rm_offset(int64_t ray, int nslab, int slab, int plane, int c) { return (ray * nslab + slab) * (int64_t)kSlabB + plane * 64 + c * 16; }

// LDS image of a staged slab (both layouts): row r (weight rows first, then the tile's rays) at r * kPRow: [plane h 64 B | plane l 64 B]
SDG_DL unsigned lds_offset(unsigned row, unsigned plane, unsigned c) { return row * (unsigned)kPRow + plane * 64u + c * 16u; }
// operand fragment of lane l for k-step ks (16 inputs) of a 32-row block starting at LDS row row0: 16 bytes = inputs 16 ks + 8 (l >> 5) + 0..7
SDG_DL unsigned frag_offset(unsigned row0, unsigned lane, unsigned ks, unsigned plane) {
  return (row0 + (lane & 31u)) * (unsigned)kPRow + plane * 64u + ks * 32u + (lane >> 5) * 16u;
}

// Loader of the ray rows of a slab, 512 threads, pieces of 64 rays (piece jp = rays 64 jp .. 64 jp + 63 of the tile).
//   ray-major:   thread t takes chunk t & 7 (plane (t & 7) >> 2, c = t & 3) of ray 64 jp + (t >> 3)
//   chunk-major: thread t takes chunk t >> 6 of ray 64 jp + (t & 63): a wave reads 1 KB of consecutive bytes
SDG_DL unsigned load_ray(bool chunk_major, unsigned tid, unsigned jp) { return 64u * jp + (chunk_major ? (tid & 63u) : (tid >> 3)); }
SDG_DL unsigned load_chunk8(bool chunk_major, unsigned tid) { return chunk_major ? (tid >> 6) : (tid & 7u); }
// chunk-major source: byte offset from the (tile's first granule, slab) base; granule_stride = slabs per ray of the segment * kGranSlab
SDG_DL unsigned cm_src_offset(unsigned ray_in_tile, unsigned chunk8_bytes, unsigned granule_stride) {
  return (ray_in_tile >> 7) * granule_stride + (ray_in_tile & 127u) * 16u + chunk8_bytes;
}

#undef SDG_DL
}  // namespace dl
}  // namespace sdg
