// sweep_plan.h -- how a batch of images is cut into launches of the select sweep and, since round 5, how the images of a launch are PACKED into
// 256-token tiles (host-side arithmetic; shared with the CPU test-suite through hostcheck.cpp).
#pragma once
#include <vector>

namespace sdg {

// Slots of the NEXT launch when `left` slots of a batch remain: launches of `cap`, the last one taking what is left, up to cap + cap / 2 (a launch of one
// or two slots would pull every key tile from HBM for itself).  cap <= 0: everything in one launch.  Why a cap at all: sixdgs_select_sweep, score.hip.
inline int sweep_launch_images(int left, int cap) {
  if (cap <= 0) return left;
  const int tail = cap + cap / 2;
  return left > tail ? cap : left;
}

// ---- token packing (round 5) -----------------------------------------------------------------------------------------------------------------------
// A workgroup of the sweep scores a tile of 256 token rows x 256 rays; its four wave rows own 64 token rows ("quarters") each and a wave row without
// tokens skips its matrix work.  An image that keeps T tokens (masked Tanks&Temples / Blender views: reference backbone.py:86-114 keeps 56-176 of 256;
// identification_module.py:80-82 scores only those) needs q = ceil(T / 64) quarters; alone in a tile it leaves 4 - q wave rows idle while all
// eight waves still move the tile's key slabs (round 4: 64 tokens cost 0.47 of 256, ideal 0.25).  So the images of a launch share tiles: a SLOT is one
// 256-row tile, and the QUARTERS of the launch's images are laid into the slots one after the other -- four views of <= 64 tokens in one tile, two of
// <= 128, and a view of 129-192 tokens takes three quarters wherever they fall, even across two tiles: every slot but the last is full.
// That is possible because nothing ties an image's quarters together inside the sweep: a wave row accumulates its own 64 tokens (one power-of-two q
// scale per QUARTER, per-lane accumulators, the 64-token butterfly stays inside a wave) and leaves its own row of partial sums; U of an image is the
// sum of its quarters' rows, formed afterwards (k_sel_finish_slots) in quarter order.  So an image's U, per-token sums and statistics are the same
// bits wherever its quarters sit and whoever their neighbours are.
constexpr int kSlotQuarters = 4;
constexpr int kSweepMaxSlots = 32;                            // slots of one launch: the table's size (the default cap 8 + tail gives <= 12)
constexpr int kSweepMaxLaunchImages = kSweepMaxSlots * kSlotQuarters;

// The table of ONE launch, passed BY VALUE to the small kernels around the sweep (q planes per slot, per-slot ctok, the merge of the statistics and the
// finish of U per image): no H2D copy, capturable in a hipGraph.  Image numbers are indices into the caller's batch.
struct SweepSlots {
  int n_slots, n_images;
  short q_img[kSweepMaxSlots][kSlotQuarters];          // image whose quarter sits in (slot, tile quarter), or -1
  short q_lq[kSweepMaxSlots][kSlotQuarters];           // which quarter of that image (token rows 64 lq .. 64 lq + 63)
  short img[kSweepMaxLaunchImages];                    // the launch's images ...
  short img_nq[kSweepMaxLaunchImages];                 // ... their number of quarters (0: an image without tokens sits in no slot) ...
  short img_q[kSweepMaxLaunchImages][kSlotQuarters];   // ... and where quarter y went: 4 * slot + tile quarter (= the row of the partial sums), or -1
};

inline int quarters_of(int n_tok) {
  const int t = n_tok < 0 ? 0 : (n_tok > 256 ? 256 : n_tok);
  return (t + 63) >> 6;
}

// Cuts a batch into launches.  h_n_tok == nullptr: token counts unknown on the host -- every image takes a whole slot (the round-4 behaviour).
// Launch by launch: the slots of the next launch from the quarters still to place (sweep_launch_images over ceil(quarters / 4), at most kSweepMaxSlots);
// then the remaining images in the caller's order, each taken if ALL its quarters still fit the launch (an image never straddles two launches: its U is
// finished from one launch's partial sums), its quarters laid down one after the other.  Images without tokens get no quarter and ride along with the
// last launch (n_slots may be 0: a launch of post-processing only).
inline std::vector<SweepSlots> sweep_pack(const int* h_n_tok, int batch, int cap) {
  auto blank = [] {
    SweepSlots t;
    t.n_slots = t.n_images = 0;
    for (int s = 0; s < kSweepMaxSlots; ++s)
      for (int w = 0; w < kSlotQuarters; ++w) t.q_img[s][w] = t.q_lq[s][w] = -1;
    for (int i = 0; i < kSweepMaxLaunchImages; ++i) {
      t.img[i] = -1, t.img_nq[i] = 0;
      for (int w = 0; w < kSlotQuarters; ++w) t.img_q[i][w] = -1;
    }
    return t;
  };
  int lcap = cap <= 0 ? kSweepMaxSlots : cap;                                   // (cap <= 0, "no cap": as many slots as the table holds)
  if (lcap + lcap / 2 > kSweepMaxSlots) lcap = kSweepMaxSlots * 2 / 3;
  std::vector<int> nq(batch > 0 ? batch : 0), left, empty;
  int qleft = 0;
  for (int i = 0; i < batch; ++i) {
    nq[i] = h_n_tok ? quarters_of(h_n_tok[i]) : kSlotQuarters;
    if (nq[i] == 0) empty.push_back(i);
    else left.push_back(i), qleft += nq[i];
  }
  std::vector<SweepSlots> out;
  while (!left.empty()) {
    int ns = sweep_launch_images((qleft + kSlotQuarters - 1) / kSlotQuarters, lcap);
    if (ns > kSweepMaxSlots) ns = kSweepMaxSlots;
    const int room = ns * kSlotQuarters;
    SweepSlots t = blank();
    int used = 0;
    std::vector<int> rest;
    for (int i : left) {
      if (used + nq[i] > room) { rest.push_back(i); continue; }
      const int k = t.n_images++;
      t.img[k] = (short)i, t.img_nq[k] = (short)nq[i];
      for (int y = 0; y < nq[i]; ++y, ++used) {
        t.q_img[used >> 2][used & 3] = (short)i, t.q_lq[used >> 2][used & 3] = (short)y;
        t.img_q[k][y] = (short)used;
      }
      qleft -= nq[i];
    }
    t.n_slots = (used + kSlotQuarters - 1) / kSlotQuarters;
    out.push_back(t);
    left.swap(rest);
  }
  for (size_t e0 = 0; e0 < empty.size();) {        // images without tokens: U = 0, no sums -- post-processing only
    if (out.empty() || out.back().n_images == kSweepMaxLaunchImages) out.push_back(blank());
    SweepSlots& t = out.back();
    while (e0 < empty.size() && t.n_images < kSweepMaxLaunchImages) t.img[t.n_images++] = (short)empty[e0++];
  }
  return out;
}

}  // namespace sdg
