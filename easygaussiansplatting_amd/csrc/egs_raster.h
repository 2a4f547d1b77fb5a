// Internal interfaces between the translation units of the rasterizer (gfx950 / CDNA4, wave64):
//   egs_sort.hip      stable LSD radix sort of (u32 key, u32 value) pairs, exclusive prefix sum
//   egs_bin.hip       getRects / createKeys / getRanges: binning records, offsets, emission, tile ranges
//   egs_draw.hip      dispatch order of the tiles, k_draw (forward blend), k_draw_bwd (backward blend)
//   egs_segments.hip  long tile lists split over several waves: plan, segment forward launches, workspace
//   egs_splat.hip     host orchestration of splat / splatB and their C-ABI entry points
// What the reference does in gsplatcu/gausplat.cu:24-159 + kernel.cu:13-271, 809-950 and how this differs by design:
//   * reference: expand (tile<<32|depth_mm) keys for all P patches, one 64-bit thrust sort of P pairs.  Here: sort
//     the N Gaussians by depth key (32-bit keys, N pairs), expand patches in that order, then one or two stable
//     passes over P on the tile id only, the id's bits spread evenly over the passes (13 bits = 7 + 6).  Same final
//     order (ties in Gaussian-index order), ~2x less sort traffic at P/N ~ 4.
//   * reference draw: 256 threads per 16x16 tile, block barrier + vote per Gaussian, 4 separate gathers per entry.
//     Here: ONE wave64 per tile, 4 pixels per lane, 64-entry chunks of packed 48-B records staged in LDS and read
//     back as wave-uniform (broadcast) ds_read_b128 -- no cross-wave barrier in the blend loop, early exit by a wave
//     vote.
//   * reference drawB: 9 same-address float atomics per (pixel, Gaussian).  Here: 4 pixels summed in-lane, four
//     entries reduced together by a transposing wave reduction (permlane swaps + DPP row merges) that leaves the 9
//     sums of an entry in 9 different lanes, which issue ONE packed atomic instruction per (tile, Gaussian) into a
//     12-float row: 256x fewer atomics.
#pragma once
#include "egs_gaussian_math.h"

namespace egs {

// ---- egs_sort.hip -------------------------------------------------------------------------------------------
struct SortWs {
  uint32_t* hist;      // [workgroup][digit]
  uint32_t* sup;       // [pass (<= 4)][superblock][digit]: must be ZERO when the sort's first kernel starts
  size_t sup_words;    // words of `sup` (what the caller zeroes)
  int nblocks;
};
size_t sort_ws_bytes(int64_t n);
bool sort_ws_carve(Carver& cv, int64_t n, SortWs* w);
int sort_passes(int begin_bit, int end_bit);
// enqueue all passes; the result ends in (keys, vals) if the pass count is even.  gsrc / gdst / cdst: the last pass
// also gathers gdst[j] = gsrc[value of the j-th item of the sorted sequence] and cdst[j] = cr_count(gdst[j])
int radix_sort(int64_t n, uint32_t* keys, uint32_t* vals, uint32_t* keys_alt, uint32_t* vals_alt, int begin_bit,
               int end_bit, const SortWs& w, hipStream_t s, const uint32_t* maxkey = nullptr,
               const uint32_t* n_dev = nullptr, uint32_t* mk_parts = nullptr, int nparts = 0,
               uint32_t* mk_out = nullptr, uint32_t* mk_host = nullptr, const uint4* gsrc = nullptr,
               uint4* gdst = nullptr, uint32_t* cdst = nullptr);
constexpr int SC_IPT = 8;             // items per thread of the scan kernels
constexpr int SC_TILE = 256 * SC_IPT;
size_t scan_ws_bytes(int64_t n);

// ---- egs_bin.hip --------------------------------------------------------------------------------------------
struct BinLayout {
  uint4 *cr, *cr_sorted;         // compact bin records in Gaussian order / in depth order
  uint32_t* cnt_sorted;          // patch counts in depth order (written next to cr_sorted)
  BinRec* br;                    // full footprint records (Gaussian order; written for big cullable rects only)
  uint32_t *dkeys, *dkeys_alt, *ids, *ids_alt, *offsets, *scan_partials, *maxkey;
  SortWs sort;
};
size_t bin_ws_bytes(int n);
bool bin_carve(void* ws, size_t bytes, int n, BinLayout* L);
// createKeys in depth order (k_bin_emit): tile keys + list values of all patches, tile ranges initialised, the
// superblock sums of the tile sort zeroed on the side
int bin_emit(int n, int gx, const BinLayout& B, uint32_t* tkeys, uint32_t* gsid, uint32_t cap, int32_t* ranges,
             int n_ranges, int with_masks, uint32_t* sort_sup, uint32_t sort_sup_words, hipStream_t s);
// getRanges over the sorted tile keys; masked / plain (nullable pair): the list without its block masks on the way
// zero / zero_words (nullable): words the kernel clears on the side (the segment plan's bins and counters)
// walk_word / walk_host (nullable pair): the longest walk the PREVIOUS render on this stream gathered in the caller's
// persistent device word is published into walk_host[1] (page-locked) and the word reset to -1 (walk_raise)
int tile_ranges(int64_t P, const uint32_t* tkeys, int32_t* ranges, const uint32_t* n_dev, const uint32_t* masked,
                int32_t* plain, hipStream_t s, int32_t* zero = nullptr, int zero_words = 0, int32_t* walk_word = nullptr,
                uint32_t* walk_host = nullptr);
// the packed 48-B records of the draw kernels from the four tensors of the op surface (+ content stamps, nullable)
int pack_records(int n, int width, int height, int footprint, float alpha_skip, const float* us, const float* cinv2ds,
                 const float* alphas, const float* colors, const int32_t* areas, float4* rec, uint32_t* stamp,
                 const uint32_t* stamp_ref, uint8_t* same, hipStream_t s);

// ---- egs_draw.hip -------------------------------------------------------------------------------------------
struct DrawParams {
  int W, H, gx, gy, T;
  float alpha_skip, tau_stop;
  float lskip;   // log2(alpha_skip), -inf when there is no skip test
  int maha_floor, alpha_clamp;
  int nan_blend;  // EgsPolicy.nan_maha == 0: an entry whose conic or centre holds a NaN blends at min(0.99, alpha) everywhere
  int map_mode;  // 0: tile = block; 1: contiguous band per XCD; 2: tile rows interleaved over XCDs
  // longest-list-first dispatch (k_tile_order): workgroup b draws tile order[b] (-1: padding) when set
  const int32_t* order;
  int ngrid;     // entries of `order` (= workgroups launched)
  // k_draw only: buffer its workgroups zero on the side (the packed gradient records of the coming backward
  // pass: 48 N bytes; the kernel is VALU-bound and leaves the memory system idle, a separate fill costs 8 us)
  float4* zero_buf;
  uint32_t zero_n4, zero_per;   // float4s in all / per workgroup
  // k_draw only (nullable): per-tile work measure for the backward pass's dispatch order -- how far the tile
  // actually walked its list (early termination makes that 0.6 .. 1.0 of the list length, tile by tile)
  int32_t* work_out;
  int32_t* walk_out;   // nullable, next to work_out: the largest contributor index of the tile (how far it was walked)
  // k_draw only (nullable, with work_out): the caller's persistent device word in which the waves gather the longest walk
  // of the render; the range kernel of the NEXT render on the stream publishes it (walk_raise, egs_draw_device.h)
  int32_t* walk_max;
  // the list values carry the tile's 4-bit block mask in their high bits (culled lists of the fused path, k_bin_emit):
  // the kernels take it from there instead of testing the record's certain-miss box per entry
  int masked;
};
DrawParams make_draw_params(int W, int H, const EgsPolicy* pol, bool backward = false);
int draw_grid(const DrawParams& p);
// capacity of a dispatch-order buffer (the per-XCD modes pad every class to the largest one)
int tile_order_len(int gx, int gy);
constexpr int TILE_ORDER_MAX_T = (16 + 24) * 1024;   // what k_tile_order handles (larger images keep the plain tile map)
int tile_order_mode(int which);                       // 0 forward, 1 backward
int tile_order_enqueue(DrawParams& p, int which, int32_t* buf, size_t buf_len, const int32_t* ranges, hipStream_t s,
                       const int32_t* work = nullptr, const int32_t* walk = nullptr, uint32_t* hint = nullptr);
// the per-tile work measure of k_draw rebuilt from `contrib` (work nullable: the walk alone)
int tile_work_from_contrib(const DrawParams& p, const int32_t* contrib, int32_t* work, int32_t* walk, hipStream_t s);
int launch_draw(const DrawParams& dp, const EgsPolicy* pol, int32_t* ranges, const int32_t* gsid, const float4* rec,
                float* image, int32_t* contrib, float* final_tau, hipStream_t s);

// ---- long lists split over several waves (egs_segments.hip) ----------------------------------------
constexpr int SEG_HDR = 16;          // header words
constexpr int SEG_SLOT_FLOATS = 256 * 6;
constexpr uint32_t SEG_TILE_MASK = 0x7FFFFu, SEG_SEG_MASK = 0x7FFu;   // item = tile | seg << 19 | kind << 30
constexpr int SEG_SPEC = 1, SEG_COMPOSE = 2;   // (0: a DIRECT item, the bare tile index)
enum { SH_ITEMS1 = 0, SH_ITEMS3 = 1, SH_SLOTS = 2, SH_MAXLEN = 3, SH_SPLIT = 4, SH_L = 6, SH_MIN = 7,
       SH_MAXWALK = 8 /* longest walk of THIS render, gathered by the draw items */,
       // the plan's global counters (k_seg_plan_count; zeroed with the bins in front of it) and the compose launch's ticket
       SH_P_SLOTS = 10, SH_P_N3 = 11 /* one 8-byte aligned pair: a single 64-bit atomic hands out both */,
       SH_P_MAXLEN = 12, SH_P_MAXWALK = 13 };
constexpr int SEG_PLAN_BINS = 4096;  // global bins of the plan's counting sort, behind the header and its 48 spare words
constexpr int SEG_PLAN_WORDS = SEG_HDR + 48 + SEG_PLAN_BINS;   // what must be ZERO when k_seg_plan_count starts
struct SegArgs {
  int32_t* hdr;        // SEG_HDR words
  int32_t* seg_base;   // [T] first state slot of a split tile, -1: not split
  int32_t* walk;       // [T] how far the forward pass walked the tile (largest contributor index; rebuilding: given)
  int32_t* items1;     // DIRECT / SPEC items: forward launches 0 and 1, and the backward launch (COMPOSE appends the
                       // segments it had to walk itself, so every walked segment of a split tile is in the list)
  int32_t* items3;     // forward, launch 2: COMPOSE(nspec), one per split tile
  int32_t* tmp;        // [T] plan scratch: (bin, rank inside the bin) of the tile's items
  int32_t* tmp2;       // [T] plan scratch: the tile's item count (-1: one DIRECT item)
  float4* st4;         // [slot][256] (C_local.rgb, tau_local) -> after COMPOSE (G.rgb, T_end)
  float* st1;          // [slot][256] last contributor (int bits) -> T_end
  float* st2;          // [slot][256] tau_local again, dense (launch 1 multiplies the taus in front of its segment)
  int slot_cap, item_cap;
  int32_t* hist_walk;  // nullable: the camera's own walk array (the NEXT render's prediction)
  int rebuild;         // splatB without the forward pass's states (egs_splat_bwd_seg): `walk` is given (from `contrib`), a
                       // tile's list ENDS there, the forward launches only rebuild the segment-end states
  int32_t* walk_max;   // nullable: the caller's persistent word the draw items gather the render's longest walk in
};
struct SegConfig { int L, split_min; };   // segment length (a power of two >= 64), shortest list that is split
SegConfig seg_config();                   // the process-wide default (egs_seg_config / EGS_SEG_L / EGS_SEG_MIN)
size_t seg_ws_bytes_for(int64_t slots, int T);
bool seg_carve(void* ws, size_t bytes, int T, SegArgs* a);
// upper bound of the work items of a render: tiles + segments
int64_t seg_item_bound(int T, int64_t patches, const SegConfig& c);
// plan + the three forward launches (+ the report to the host's hint words) over a carved workspace
// plan_zeroed: the SEG_PLAN_WORDS words at sga.hdr are already zero (tile_ranges cleared them on the side)
// walk_word (nullable): the caller's persistent word for the render's longest walk (see tile_ranges)
int draw_segments_forward(DrawParams& dp, const EgsPolicy* pol, SegArgs& sga, const SegConfig& cfg, int64_t patches,
                          const int32_t* hist, int speculate, bool fix_pass, int32_t* walk_word, uint32_t* seg_hint,
                          bool plan_zeroed,
                          int32_t* ranges, const int32_t* gsid, const float4* rec, float* image, int32_t* contrib,
                          float* final_tau, hipStream_t s);
int launch_draw_bwd(const DrawParams& dp, const EgsPolicy* pol, const int32_t* ranges, const int32_t* gsid,
                    const float4* rec, const float* final_tau, const int32_t* contrib, const float* dLdg, float* gpack,
                    hipStream_t s);
// the same over the forward pass's work items (one wave per segment of a split tile)
int launch_draw_bwd_seg(const DrawParams& dp, const EgsPolicy* pol, const int32_t* ranges, const int32_t* gsid,
                        const float4* rec, const float* final_tau, const int32_t* contrib, const float* dLdg,
                        float* gpack, const SegArgs& sga, int grid, hipStream_t s);
int unpack_grads(int n, const float* gpack, float* dus, float* dcinv, float* dalpha, float* dcolor, hipStream_t s);

}  // namespace egs
