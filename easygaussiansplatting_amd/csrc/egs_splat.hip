// Host orchestration of splat / splatB (reference gsplatcu/gausplat.cu:24-159) and their C-ABI entry points
// (include/egs_hip.h): emission + tile sort + ranges + draw, and the backward draw into packed gradient records.
// The kernels live in egs_sort.hip, egs_bin.hip, egs_draw.hip and egs_segments.hip (interfaces: egs_raster.h).
#include "egs_raster.h"

#include <algorithm>

namespace egs {

static int tile_bits(int T) {
  int b = 1;
  while ((1 << b) < T) ++b;
  return b;
}

struct DrawLayout {
  uint32_t *tkeys, *tkeys_alt, *gsid_alt;
  float4* rec;
  int32_t* order;   // dispatch order of the tiles (k_tile_order)
  SortWs sort;
};
static size_t draw_ws_bytes(int n, int64_t P, int width, int height) {
  const size_t N = (size_t)(n > 0 ? n : 1), PP = (size_t)(P > 0 ? P : 1);
  const size_t ord = (size_t)tile_order_len(div_up(width, EGS_TILE), div_up(height, EGS_TILE));
  return 3 * align_up(PP * 4, 256) + align_up(N * 48, 256) + align_up(ord * 4, 256) + sort_ws_bytes(P) + 4096;
}
static bool draw_carve(void* ws, size_t bytes, int n, int64_t P, int width, int height, DrawLayout* L) {
  Carver cv(ws, bytes);
  const size_t N = (size_t)(n > 0 ? n : 1), PP = (size_t)(P > 0 ? P : 1);
  L->tkeys = cv.take<uint32_t>(PP);
  L->tkeys_alt = cv.take<uint32_t>(PP);
  L->gsid_alt = cv.take<uint32_t>(PP);
  L->rec = cv.take<float4>(3 * N);
  L->order = cv.take<int32_t>((size_t)tile_order_len(div_up(width, EGS_TILE), div_up(height, EGS_TILE)));
  return sort_ws_carve(cv, P, &L->sort) && cv.ok();
}

// [records | packed gradients | tile dispatch order (bounded: larger images keep the plain tile map)]
constexpr size_t BWD_ORDER_CAP = (size_t)1 << 18;   // tiles: up to 8192 x 8192 pixels

int splat_bwd_packed(int n, int64_t patches, int width, int height, const float* us, const float* cinv2ds,
                     const float* alphas, const float* colors, const int32_t* areas, const EgsPolicy* pol,
                     const int32_t* contrib, const float* final_tau, const int32_t* patch_range_per_tile,
                     const int32_t* gsid_per_patch, const float* dloss_dgammas, void* ws, size_t ws_bytes,
                     float** gpack_out, void* stream, const void* rec_in, const int32_t* tile_order,
                     float* grad_records, bool keep_forward_order, bool masked_lists, void* seg_ws,
                     size_t seg_ws_bytes, int rebuild, uint32_t* seg_hint) {
  // rebuild != 0 (with seg_ws of egs_seg_rebuild_ws_bytes): no forward pass left its segment states here -- the public
  // splatB is handed tensors only -- so they are REBUILT first: every tile's walk from `contrib`, then the forward
  // segment launches over [0, walk) with their pixels going to scratch.  seg_hint (nullable): the page-locked words
  // that learn the longest walk (both paths report it: a host decides the path of its NEXT call from it).
  hipStream_t s = (hipStream_t)stream;
  const float4* rec = rec_in ? (const float4*)rec_in : (const float4*)ws;
  // [N][12] packed gradient records: the caller's (already zeroed by the forward draw kernel) or a piece of ws
  float* gpack = grad_records ? grad_records : (float*)((char*)ws + align_up((size_t)n * 48, 256));
  *gpack_out = gpack;
  (void)ws_bytes;
  if (!grad_records) EGS_HIP(hipMemsetAsync(gpack, 0, (size_t)n * 48, s));
  if (patches == 0) return 0;
  EGS_CHECK_ARG(contrib && final_tau && patch_range_per_tile && gsid_per_patch && dloss_dgammas);
  EGS_CHECK_ARG(rec_in || (us && cinv2ds && alphas && colors && (areas || pol->footprint != 1)));
  EGS_CHECK_ARG(rec_in || (us && alphas && colors && (pol->footprint == 0 || areas)));
  DrawParams dp = make_draw_params(width, height, pol, true);
  dp.masked = (masked_lists && pol->footprint == 0 && pol->alpha_skip > 0.f) ? 1 : 0;
  if (!rec_in) {
    const int rc = pack_records(n, width, height, pol->footprint, pol->alpha_skip, us, cinv2ds, alphas, colors, areas,
                                (float4*)ws, nullptr, nullptr, nullptr, s);
    if (rc) return rc;
  }
  if (seg_ws) {   // the forward pass split its long lists (egs_splat_draw_rec_seg): one wave per segment
    SegArgs sga;
    const size_t hw = (size_t)width * height;
    const size_t scratch = rebuild ? align_up(20 * hw, 256) + 256 : 0;
    if (pol->footprint != 0 || !(pol->alpha_skip > 0.f) || !(pol->tau_stop > 0.f) || seg_ws_bytes <= scratch ||
        !seg_carve(seg_ws, seg_ws_bytes - scratch, dp.T, &sga)) {
      set_error(EGS_ERR_WORKSPACE, "segment workspace too small (or a policy without a skip / stop threshold)", __FILE__, __LINE__);
      return EGS_ERR_WORKSPACE;
    }
    // (the grid covers the workspace's item capacity -- tiles + state slots, what egs_seg_ws_bytes sized it for -- not a
    // bound formed from the CURRENT egs_seg_config: the render's own L is in the workspace header)
    const int grid = sga.item_cap;
    if (rebuild) {
      sga.rebuild = 1;
      char* sc = (char*)(((uintptr_t)seg_ws + seg_ws_bytes - scratch + 255) & ~(uintptr_t)255);
      float* simg = (float*)sc;
      int32_t* scont = (int32_t*)(sc + 12 * hw);
      float* stau = (float*)(sc + 16 * hw);
      DrawParams fp = make_draw_params(width, height, pol);
      fp.masked = dp.masked;
      int32_t* rg = const_cast<int32_t*>(patch_range_per_tile);   // (only DIRECT items of an empty tile write it: none here)
      int rc = tile_work_from_contrib(dp, contrib, nullptr, sga.walk, s);
      if (rc) return rc;
      rc = draw_segments_forward(fp, pol, sga, seg_config(), patches, (const int32_t*)sga.walk, 0, true, nullptr, seg_hint,
                                 false, rg, gsid_per_patch, rec, simg, scont, stau, s);
      if (rc) return rc;
    }
    return launch_draw_bwd_seg(dp, pol, patch_range_per_tile, gsid_per_patch, rec, final_tau, contrib, dloss_dgammas,
                               gpack, sga, grid, s);
  }
  static const int by_work = [] { const char* e = getenv("EGS_DRAWB_BY_WORK"); return e ? atoi(e) : 1; }();
  const bool same_mode = tile_order_mode(0) == tile_order_mode(1) && tile_order_mode(1) > 0;
  if (tile_order && keep_forward_order && same_mode) {
    // the forward pass already dispatched by measured work (that of the camera's previous render, one step or
    // one epoch old -- as good a key for this pass as for that one): no second k_tile_order (8 us)
    dp.order = tile_order;
    dp.ngrid = tile_order_mode(1) >= 3 ? tile_order_len(dp.gx, dp.gy) : dp.T;
  } else if (tile_order && by_work && tile_order_mode(1) > 0 && (size_t)tile_order_len(dp.gx, dp.gy) <= BWD_ORDER_CAP) {
    // the forward draw kernel left behind how far every tile walked its list: order the tiles by THAT (the list
    // length mis-ranks tiles whose pixels saturate early; simulated with the measured work of the 1 M scene:
    // makespan 1.11 x ideal by length, 1.03 x by work)
    int32_t* order = (int32_t*)((char*)ws + 2 * align_up((size_t)n * 48, 256));
    const int32_t* wk = tile_order + tile_order_len(dp.gx, dp.gy);      // [work | walk] of the forward draw
    const int rc = tile_order_enqueue(dp, 1, order, BWD_ORDER_CAP, patch_range_per_tile, s, wk,
                                      seg_hint ? wk + dp.T : nullptr, seg_hint);
    if (rc) return rc;
  } else if (tile_order && same_mode) {
    // the forward pass left its dispatch order behind (same mode): no second k_tile_order
    dp.order = tile_order;
    dp.ngrid = tile_order_mode(1) >= 3 ? tile_order_len(dp.gx, dp.gy) : dp.T;
  } else {
    // no record of the forward pass (the seven-op surface: splatB only gets tensors): the work measure is rebuilt
    // from `contrib`, exactly as k_draw would have left it, and the tiles are ordered by it (k_draw_bwd 465 ->
    // 445 us against ordering by list length, for a 4-us kernel)
    int32_t* order = (int32_t*)((char*)ws + 2 * align_up((size_t)n * 48, 256));
    const size_t len = (size_t)tile_order_len(dp.gx, dp.gy);
    int32_t* work = nullptr;
    int32_t* walk = nullptr;
    if (by_work && tile_order_mode(1) > 0 && len + (size_t)dp.T <= BWD_ORDER_CAP) {
      work = order + len;
      if (seg_hint && len + 2 * (size_t)dp.T <= BWD_ORDER_CAP) walk = work + dp.T;
      const int rc = tile_work_from_contrib(dp, contrib, work, walk, s);
      if (rc) return rc;
    }
    const int rc = tile_order_enqueue(dp, 1, order, BWD_ORDER_CAP, patch_range_per_tile, s, work, walk,
                                      walk ? seg_hint : nullptr);
    if (rc) return rc;
  }
  return launch_draw_bwd(dp, pol, patch_range_per_tile, gsid_per_patch, rec, final_tau, contrib, dloss_dgammas, gpack, s);
}
}  // namespace egs

using namespace egs;

extern "C" size_t egs_splat_draw_ws_bytes(int n, int64_t patches, int width, int height) {
  return draw_ws_bytes(n, patches, width, height);
}

static int splat_draw_impl(int n, int64_t patches, int width, int height, const float* us,
                           const float* cinv2ds, const float* alphas, const float* colors,
                           const int32_t* areas, const EgsPolicy* pol, const void* ws_bin, void* ws_draw,
                           size_t ws_draw_bytes, const float4* rec_in, float* image, int32_t* contrib,
                           float* final_tau, int32_t* patch_range_per_tile, int32_t* gsid_per_patch,
                           void* stream, const uint32_t* patches_dev = nullptr, int32_t* tile_order = nullptr,
                           float* grad_records = nullptr, const int32_t* prev_tile_work = nullptr,
                           int order_ready = 0, int flags = 0, int32_t* gsid_plain = nullptr, void* seg_ws = nullptr,
                           size_t seg_ws_bytes = 0, uint32_t* seg_hint = nullptr, int32_t* walk_word = nullptr) {
  // walk_word (nullable, with seg_hint): the caller's PERSISTENT device word (one per problem size and stream, -1 before
  // its first use) in which this render's draw items gather its longest walk; the range kernel of this call first
  // publishes what the previous render left there into seg_hint[1]
  // seg_ws != NULL (egs_seg_ws_bytes): long lists are split over several waves (k_draw_seg; the backward pass then
  // takes the same workspace); flags & EGS_DRAW_SEG_HISTORY: the walk part of tile_order holds what an earlier render of
  // this camera measured.  seg_hint (nullable, page-locked): receives the longest list of this render.
  // gsid_plain (nullable, with EGS_DRAW_MASKED_LISTS): receives the list values without their masks
  // flags & EGS_DRAW_CULLED_LISTS: the binning stage counted the footprint-culled tiles (egs_fused_forward with
  // cull_lists): the lists are emitted with block masks in the high bits of their values and drawn from those
  // order_ready != 0: tile_order already holds a dispatch order (an earlier render through the SAME buffer left
  // it there): it is used as it stands, no k_tile_order launch; the work part is still rewritten by the draw
  // prev_tile_work != NULL (T ints): the work the draw kernel measured per tile the LAST time this camera was
  // rendered -- a much better sort key for the dispatch order than the list length (pixels saturate)
  // grad_records != NULL ([N][12] floats): zeroed on the side by the draw kernel for the coming backward pass
  // tile_order != NULL (egs_tile_order_len ints): the dispatch order of the tiles is written there, for the
  // backward pass to reuse (otherwise it lives in ws_draw and the backward pass computes its own)
  // patches_dev != NULL: `patches` is only the capacity of gsid_per_patch / ws_draw, the real count is read on
  // the device (the host has not seen it yet)
  EGS_CHECK_ARG(n >= 0 && patches >= 0 && patches < (int64_t)0x7FFFFFFF && width > 0 && height > 0 && pol);
  EGS_CHECK_ARG(image && contrib && final_tau && patch_range_per_tile);
  hipStream_t s = (hipStream_t)stream;
  DrawParams dp = make_draw_params(width, height, pol);
  const bool masked = (flags & (EGS_DRAW_CULLED_LISTS | EGS_DRAW_MASKED_LISTS)) && pol->footprint == 0 &&
                      pol->alpha_skip > 0.f;
  EGS_CHECK_ARG(!masked || n < (1 << EGS_GSID_BITS));
  dp.masked = masked ? 1 : 0;
  if (grad_records && n > 0 && (patches == 0)) EGS_HIP(hipMemsetAsync(grad_records, 0, (size_t)n * 48, s));
  if (n == 0 || patches == 0) {  // nothing to draw: all outputs are zero
    const size_t hw = (size_t)width * height;
    EGS_HIP(hipMemsetAsync(patch_range_per_tile, 0, (size_t)dp.T * 8, s));
    EGS_HIP(hipMemsetAsync(image, 0, 12 * hw, s));
    EGS_HIP(hipMemsetAsync(contrib, 0, 4 * hw, s));
    EGS_HIP(hipMemsetAsync(final_tau, 0, 4 * hw, s));
    if (tile_order) {
      // the caller keeps [order | work] between renders and will trust it next time (order_ready): it must hold
      // a valid permutation and the work of THIS render (none) whatever happened here
      const size_t olen = (size_t)tile_order_len(dp.gx, dp.gy);
      if (!order_ready) {
        const int rc = tile_order_enqueue(dp, 0, tile_order, olen, patch_range_per_tile, s, nullptr);
        if (rc) return rc;
      }
      EGS_HIP(hipMemsetAsync(tile_order + olen, 0, (size_t)dp.T * 8, s));   // work and walk
    }
    if (seg_ws) {   // a backward pass may still be handed the workspace: no items, nothing split
      SegArgs sa;
      if (seg_carve(seg_ws, seg_ws_bytes, dp.T, &sa)) EGS_HIP(hipMemsetAsync(sa.hdr, 0, SEG_HDR * 4, s));
    }
    return 0;
  }
  EGS_CHECK_ARG(ws_bin && ws_draw && gsid_per_patch);
  EGS_CHECK_ARG(rec_in || (us && cinv2ds && alphas && colors && areas));
  BinLayout B;
  if (!bin_carve(const_cast<void*>(ws_bin), bin_ws_bytes(n), n, &B)) return EGS_ERR_WORKSPACE;
  DrawLayout D;
  if (!draw_carve(ws_draw, ws_draw_bytes, n, patches, width, height, &D)) {
    set_error(EGS_ERR_WORKSPACE, "draw workspace too small", __FILE__, __LINE__);
    return EGS_ERR_WORKSPACE;
  }
  const int tb = tile_bits(dp.T);
  const int passes = sort_passes(0, tb);
  uint32_t* gs_primary = (uint32_t*)gsid_per_patch;
  // choose the emission buffers so that the sorted result lands in the primary ones
  uint32_t* k0 = (passes & 1) ? D.tkeys_alt : D.tkeys;
  uint32_t* k1 = (passes & 1) ? D.tkeys : D.tkeys_alt;
  uint32_t* v0 = (passes & 1) ? D.gsid_alt : gs_primary;
  uint32_t* v1 = (passes & 1) ? gs_primary : D.gsid_alt;
  int rc = bin_emit(n, dp.gx, B, k0, v0, (uint32_t)patches, patch_range_per_tile, 2 * dp.T, dp.masked, D.sort.sup,
                    (uint32_t)D.sort.sup_words, s);
  if (rc) return rc;
  const float4* rec = rec_in ? rec_in : D.rec;
  if (!rec_in) {
    rc = pack_records(n, width, height, pol->footprint, pol->alpha_skip, us, cinv2ds, alphas, colors, areas, D.rec,
                      nullptr, nullptr, nullptr, s);
    if (rc) return rc;
  }
  rc = radix_sort(patches, k0, v0, k1, v1, 0, tb, D.sort, s, nullptr, patches_dev);
  if (rc) return rc;
  SegArgs sga;
  const bool seg = seg_ws && pol->footprint == 0 && pol->alpha_skip > 0.f && pol->tau_stop > 0.f &&
                   dp.T <= (int)SEG_TILE_MASK && seg_carve(seg_ws, seg_ws_bytes, dp.T, &sga);
  // (the range kernel clears the bins and counters of the segment plan on the side, and publishes the longest walk of the
  // previous render on this stream)
  rc = tile_ranges(patches, D.tkeys, patch_range_per_tile, patches_dev,
                   (const uint32_t*)(gsid_plain ? gsid_per_patch : nullptr), gsid_plain, s, seg ? sga.hdr : nullptr,
                   seg ? SEG_PLAN_WORDS : 0, walk_word, seg_hint);
  if (rc) return rc;
  if (seg_ws && !seg) {
    set_error(EGS_ERR_WORKSPACE, "segment workspace too small (or a policy without a skip / stop threshold)", __FILE__, __LINE__);
    return EGS_ERR_WORKSPACE;
  }
  if (seg) {
    const SegConfig cfg = seg_config();   // (read once per render: the plan kernel leaves L in the workspace header)
    const size_t olen = (size_t)tile_order_len(dp.gx, dp.gy);
    const int32_t* hist = (tile_order && (flags & EGS_DRAW_SEG_HISTORY)) ? tile_order + olen + dp.T : nullptr;
    sga.hist_walk = tile_order ? tile_order + olen + dp.T : nullptr;
    const int speculate = (flags & EGS_DRAW_SEG_SPECULATE) ? 1 : 0;    // (with a walk on record: where that looks stale)
    if (tile_order) dp.work_out = tile_order + olen;
    if (grad_records) {   // (zero_per: set by draw_segments_forward from its grid)
      dp.zero_buf = (float4*)grad_records;
      dp.zero_n4 = (uint32_t)(3 * (size_t)n);
    }
    return draw_segments_forward(dp, pol, sga, cfg, patches, hist, speculate, hist || speculate, walk_word, seg_hint, true,
                                 patch_range_per_tile, gsid_per_patch, rec, image, contrib, final_tau, s);
  }
  if (order_ready && tile_order && tile_order_mode(0) > 0 && dp.T <= TILE_ORDER_MAX_T) {
    dp.order = tile_order;
    dp.ngrid = tile_order_mode(0) >= 3 ? tile_order_len(dp.gx, dp.gy) : dp.T;
  } else {
    // (prev_tile_work is the work part of a camera's own buffer: its walk part lies T ints behind it)
    // (no hint words from here, as in round 5 -- the walk of the camera's PREVIOUS render is stale after reset_alpha and
    // would overwrite what the range kernel just published; the draw waves gather both words of this render)
    rc = tile_order_enqueue(dp, 0, tile_order ? tile_order : D.order, (size_t)tile_order_len(dp.gx, dp.gy),
                            patch_range_per_tile, s, prev_tile_work, nullptr, walk_word ? nullptr : seg_hint);
    if (rc) return rc;
  }
  if (tile_order) { dp.work_out = tile_order + tile_order_len(dp.gx, dp.gy); dp.walk_out = dp.work_out + dp.T; }
  if (tile_order) dp.walk_max = walk_word;     // (every render refreshes the host's hint, one render late)
  if (grad_records) {
    dp.zero_buf = (float4*)grad_records;
    dp.zero_n4 = (uint32_t)(3 * (size_t)n);
    dp.zero_per = (dp.zero_n4 + (uint32_t)draw_grid(dp) - 1) / (uint32_t)draw_grid(dp);
  }
  return launch_draw(dp, pol, patch_range_per_tile, gsid_per_patch, rec, image, contrib, final_tau, s);
}

extern "C" int egs_splat_draw(int n, int64_t patches, int width, int height, const float* us,
                              const float* cinv2ds, const float* alphas, const float* colors,
                              const int32_t* areas, const EgsPolicy* pol, const void* ws_bin, void* ws_draw,
                              size_t ws_draw_bytes, float* image, int32_t* contrib, float* final_tau,
                              int32_t* patch_range_per_tile, int32_t* gsid_per_patch, void* stream) {
  return splat_draw_impl(n, patches, width, height, us, cinv2ds, alphas, colors, areas, pol, ws_bin, ws_draw,
                         ws_draw_bytes, nullptr, image, contrib, final_tau, patch_range_per_tile, gsid_per_patch,
                         stream);
}

// as egs_splat_draw, enqueued BEFORE the host has read total_patches (see egs_splat_draw_rec_dev)
extern "C" int egs_splat_draw_dev(int n, int64_t patch_capacity, const uint32_t* total_patches, int width, int height,
                                  const float* us, const float* cinv2ds, const float* alphas, const float* colors,
                                  const int32_t* areas, const EgsPolicy* pol, const void* ws_bin, void* ws_draw,
                                  size_t ws_draw_bytes, float* image, int32_t* contrib, float* final_tau,
                                  int32_t* patch_range_per_tile, int32_t* gsid_per_patch, void* stream) {
  EGS_CHECK_ARG(total_patches && patch_capacity > 0);
  return splat_draw_impl(n, patch_capacity, width, height, us, cinv2ds, alphas, colors, areas, pol, ws_bin, ws_draw,
                         ws_draw_bytes, nullptr, image, contrib, final_tau, patch_range_per_tile, gsid_per_patch,
                         stream, total_patches);
}

// as egs_splat_draw, with the packed 2D records already built (egs_fused_forward)
extern "C" int egs_splat_draw_rec(int n, int64_t patches, int width, int height, const void* rec,
                                  const EgsPolicy* pol, const void* ws_bin, void* ws_draw, size_t ws_draw_bytes,
                                  float* image, int32_t* contrib, float* final_tau,
                                  int32_t* patch_range_per_tile, int32_t* gsid_per_patch, int32_t* tile_order,
                                  float* grad_records, const int32_t* prev_tile_work, int order_ready, int flags,
                                  void* stream) {
  EGS_CHECK_ARG(rec || n == 0);
  return splat_draw_impl(n, patches, width, height, nullptr, nullptr, nullptr, nullptr, nullptr, pol, ws_bin,
                         ws_draw, ws_draw_bytes, (const float4*)rec, image, contrib, final_tau,
                         patch_range_per_tile, gsid_per_patch, stream, nullptr, tile_order, grad_records,
                         prev_tile_work, order_ready, flags);
}

// as egs_splat_draw_rec, enqueued BEFORE the host has read total_patches: patch_capacity sizes
// gsid_per_patch / ws_draw, the real count comes from total_patches[0] on the device.  When
// host_totals != NULL (page-locked host memory), total_patches[0..1] is copied there first, in stream order.
extern "C" int egs_splat_draw_rec_dev(int n, int64_t patch_capacity, const uint32_t* total_patches,
                                      uint32_t* host_totals, int width, int height, const void* rec,
                                      const EgsPolicy* pol, const void* ws_bin, void* ws_draw, size_t ws_draw_bytes,
                                      float* image, int32_t* contrib, float* final_tau,
                                      int32_t* patch_range_per_tile, int32_t* gsid_per_patch, int32_t* tile_order,
                                      float* grad_records, const int32_t* prev_tile_work, int order_ready, int flags,
                                      void* stream) {
  EGS_CHECK_ARG((rec || n == 0) && total_patches && patch_capacity > 0);
  if (host_totals)
    EGS_HIP(hipMemcpyAsync(host_totals, total_patches, 8, hipMemcpyDeviceToHost, (hipStream_t)stream));
  return splat_draw_impl(n, patch_capacity, width, height, nullptr, nullptr, nullptr, nullptr, nullptr, pol, ws_bin,
                         ws_draw, ws_draw_bytes, (const float4*)rec, image, contrib, final_tau,
                         patch_range_per_tile, gsid_per_patch, stream, total_patches, tile_order, grad_records,
                         prev_tile_work, order_ready, flags);
}

// egs_splat_draw_rec / _dev with flags = EGS_DRAW_MASKED_LISTS for the seven-op surface: gsid_per_patch receives the
// list the draw kernels walk (with masks), gsid_plain the list the CALLER of splat gets (gausplat.cu:108-111), written
// by the range kernel on its way over the sorted keys (no egs_strip_list_masks launch)
extern "C" int egs_splat_draw_rec_plain(int n, int64_t patches, int width, int height, const void* rec,
                                        const EgsPolicy* pol, const void* ws_bin, void* ws_draw, size_t ws_draw_bytes,
                                        float* image, int32_t* contrib, float* final_tau,
                                        int32_t* patch_range_per_tile, int32_t* gsid_per_patch, int32_t* gsid_plain,
                                        int32_t* tile_order, float* grad_records, int flags, void* stream) {
  EGS_CHECK_ARG(rec || n == 0);
  EGS_CHECK_ARG(!gsid_plain || ((((uintptr_t)gsid_plain | (uintptr_t)gsid_per_patch) & 15) == 0));
  return splat_draw_impl(n, patches, width, height, nullptr, nullptr, nullptr, nullptr, nullptr, pol, ws_bin,
                         ws_draw, ws_draw_bytes, (const float4*)rec, image, contrib, final_tau,
                         patch_range_per_tile, gsid_per_patch, stream, nullptr, tile_order, grad_records,
                         nullptr, 0, flags, gsid_plain);
}
extern "C" int egs_splat_draw_rec_dev_plain(int n, int64_t patch_capacity, const uint32_t* total_patches, int width,
                                            int height, const void* rec, const EgsPolicy* pol, const void* ws_bin,
                                            void* ws_draw, size_t ws_draw_bytes, float* image, int32_t* contrib,
                                            float* final_tau, int32_t* patch_range_per_tile, int32_t* gsid_per_patch,
                                            int32_t* gsid_plain, int32_t* tile_order, float* grad_records, int flags,
                                            void* stream) {
  EGS_CHECK_ARG((rec || n == 0) && total_patches && patch_capacity > 0);
  EGS_CHECK_ARG(!gsid_plain || ((((uintptr_t)gsid_plain | (uintptr_t)gsid_per_patch) & 15) == 0));
  return splat_draw_impl(n, patch_capacity, width, height, nullptr, nullptr, nullptr, nullptr, nullptr, pol, ws_bin,
                         ws_draw, ws_draw_bytes, (const float4*)rec, image, contrib, final_tau,
                         patch_range_per_tile, gsid_per_patch, stream, total_patches, tile_order, grad_records,
                         nullptr, 0, flags, gsid_plain);
}

// the draw stage of egs_splat_draw_rec / _dev (total_patches NULL: `patches` is exact) with a segment workspace
extern "C" int egs_splat_draw_rec_seg(int n, int64_t patches, const uint32_t* total_patches, int width, int height,
                                      const void* rec, const EgsPolicy* pol, const void* ws_bin, void* ws_draw,
                                      size_t ws_draw_bytes, float* image, int32_t* contrib, float* final_tau,
                                      int32_t* patch_range_per_tile, int32_t* gsid_per_patch, int32_t* tile_order,
                                      float* grad_records, const int32_t* prev_tile_work, int order_ready, int flags,
                                      void* seg_ws, size_t seg_ws_bytes, uint32_t* seg_hint, int32_t* walk_word,
                                      int32_t* gsid_plain, void* stream) {
  // gsid_plain (nullable, with EGS_DRAW_MASKED_LISTS: the seven-op surface): receives the list without its masks
  EGS_CHECK_ARG((rec || n == 0) && (!total_patches || patches > 0));
  EGS_CHECK_ARG(!gsid_plain || ((((uintptr_t)gsid_plain | (uintptr_t)gsid_per_patch) & 15) == 0));
  return splat_draw_impl(n, patches, width, height, nullptr, nullptr, nullptr, nullptr, nullptr, pol, ws_bin, ws_draw,
                         ws_draw_bytes, (const float4*)rec, image, contrib, final_tau, patch_range_per_tile,
                         gsid_per_patch, stream, total_patches, tile_order, grad_records, prev_tile_work, order_ready,
                         flags, gsid_plain, seg_ws, seg_ws_bytes, seg_hint, walk_word);
}

extern "C" size_t egs_splat_bwd_ws_bytes(int n) {
  return 2 * align_up((size_t)(n > 0 ? n : 1) * 48, 256) + BWD_ORDER_CAP * 4 + 256;
}

extern "C" size_t egs_seg_rebuild_ws_bytes(int64_t patch_capacity, int width, int height);

// splatB from the packed records (and, nullable, the [order | work] buffer the forward draw left behind: the tiles
// are then dispatched by the work that draw MEASURED, no k_tile_work pass over `contrib`)
extern "C" int egs_splat_bwd_rec(int n, int64_t patches, int width, int height, const void* rec, const EgsPolicy* pol,
                                 const int32_t* contrib, const float* final_tau, const int32_t* patch_range_per_tile,
                                 const int32_t* gsid_per_patch, const float* dloss_dgammas, void* ws, size_t ws_bytes,
                                 const int32_t* tile_order, float* grad_records, float* dloss_dus,
                                 float* dloss_dcinv2ds, float* dloss_dalphas, float* dloss_dcolors, void* stream) {
  return egs_splat_bwd_rec_lists(n, patches, width, height, rec, pol, contrib, final_tau, patch_range_per_tile,
                                 gsid_per_patch, dloss_dgammas, ws, ws_bytes, tile_order, grad_records, dloss_dus,
                                 dloss_dcinv2ds, dloss_dalphas, dloss_dcolors, 0, stream);
}

// the same; flags = EGS_DRAW_MASKED_LISTS: gsid_per_patch is the list WITH block masks the forward draw walked
// (egs_splat_bin_pack + egs_splat_draw_rec*), not the stripped copy the caller of splat got back
extern "C" int egs_splat_bwd_rec_lists(int n, int64_t patches, int width, int height, const void* rec,
                                       const EgsPolicy* pol, const int32_t* contrib, const float* final_tau,
                                       const int32_t* patch_range_per_tile, const int32_t* gsid_per_patch,
                                       const float* dloss_dgammas, void* ws, size_t ws_bytes,
                                       const int32_t* tile_order, float* grad_records, float* dloss_dus,
                                       float* dloss_dcinv2ds, float* dloss_dalphas, float* dloss_dcolors, int flags,
                                       void* stream) {
  // grad_records (nullable, [N][12] floats): the packed gradient records, ALREADY ZERO (the forward draw cleared
  // them on the side, egs_splat_draw_rec*'s grad_records): no 48 N-byte fill in front of the backward draw
  EGS_CHECK_ARG(n >= 0 && patches >= 0 && width > 0 && height > 0 && pol);
  if (n == 0) return 0;
  EGS_CHECK_ARG(rec && ws && dloss_dus && dloss_dcinv2ds && dloss_dalphas && dloss_dcolors);
  if (ws_bytes < egs_splat_bwd_ws_bytes(n)) {
    set_error(EGS_ERR_WORKSPACE, "splat_bwd workspace too small", __FILE__, __LINE__);
    return EGS_ERR_WORKSPACE;
  }
  float* gpack = nullptr;
  int rc = splat_bwd_packed(n, patches, width, height, nullptr, nullptr, nullptr, nullptr, nullptr, pol, contrib,
                            final_tau, patch_range_per_tile, gsid_per_patch, dloss_dgammas, ws, ws_bytes, &gpack, stream,
                            rec, tile_order, grad_records, false,
                            (flags & (EGS_DRAW_CULLED_LISTS | EGS_DRAW_MASKED_LISTS)) != 0);
  if (rc) return rc;
  return unpack_grads(n, gpack, dloss_dus, dloss_dcinv2ds, dloss_dalphas, dloss_dcolors, (hipStream_t)stream);
}

// splatB with everything optional that a host may or may not have: the packed records (else packed here from the four
// tensors), the [order | work | walk] buffer and the cleared gradient records of the forward draw, the forward's
// segment workspace (rebuild == 0) or a fresh one of egs_seg_rebuild_ws_bytes (rebuild != 0: the segment states are
// rebuilt from contrib / final_tau first), and the hint words.  seg_ws == NULL: the unsplit kernel.
extern "C" size_t egs_seg_rebuild_ws_bytes(int64_t patch_capacity, int width, int height) {
  return egs_seg_ws_bytes(patch_capacity, width, height) + align_up((size_t)20 * width * height, 256) + 512;
}
extern "C" int egs_splat_bwd_seg(int n, int64_t patches, int width, int height, const float* us, const float* cinv2ds,
                                 const float* alphas, const float* colors, const void* rec, const EgsPolicy* pol,
                                 const int32_t* contrib, const float* final_tau, const int32_t* patch_range_per_tile,
                                 const int32_t* gsid_per_patch, const float* dloss_dgammas, void* ws, size_t ws_bytes,
                                 const int32_t* tile_order, float* grad_records, float* dloss_dus,
                                 float* dloss_dcinv2ds, float* dloss_dalphas, float* dloss_dcolors, int flags,
                                 void* seg_ws, size_t seg_ws_bytes, int rebuild, uint32_t* seg_hint, void* stream) {
  EGS_CHECK_ARG(n >= 0 && patches >= 0 && width > 0 && height > 0 && pol);
  if (n == 0) return 0;
  EGS_CHECK_ARG(ws && dloss_dus && dloss_dcinv2ds && dloss_dalphas && dloss_dcolors);
  EGS_CHECK_ARG(rec || (us && cinv2ds && alphas && colors && pol->footprint != 1));
  if (ws_bytes < egs_splat_bwd_ws_bytes(n)) {
    set_error(EGS_ERR_WORKSPACE, "splat_bwd workspace too small", __FILE__, __LINE__);
    return EGS_ERR_WORKSPACE;
  }
  float* gpack = nullptr;
  int rc = splat_bwd_packed(n, patches, width, height, us, cinv2ds, alphas, colors, nullptr, pol, contrib, final_tau,
                            patch_range_per_tile, gsid_per_patch, dloss_dgammas, ws, ws_bytes, &gpack, stream, rec,
                            tile_order, grad_records, false,
                            (flags & (EGS_DRAW_CULLED_LISTS | EGS_DRAW_MASKED_LISTS)) != 0, seg_ws, seg_ws_bytes, rebuild,
                            seg_hint);
  if (rc) return rc;
  return unpack_grads(n, gpack, dloss_dus, dloss_dcinv2ds, dloss_dalphas, dloss_dcolors, (hipStream_t)stream);
}

extern "C" int egs_splat_bwd(int n, int64_t patches, int width, int height, const float* us,
                             const float* cinv2ds, const float* alphas, const float* colors,
                             const int32_t* areas, const EgsPolicy* pol, const int32_t* contrib,
                             const float* final_tau, const int32_t* patch_range_per_tile,
                             const int32_t* gsid_per_patch, const float* dloss_dgammas, void* ws, size_t ws_bytes,
                             float* dloss_dus, float* dloss_dcinv2ds, float* dloss_dalphas, float* dloss_dcolors,
                             void* stream) {
  EGS_CHECK_ARG(n >= 0 && patches >= 0 && width > 0 && height > 0 && pol);
  if (n == 0) return 0;
  EGS_CHECK_ARG(ws && dloss_dus && dloss_dcinv2ds && dloss_dalphas && dloss_dcolors);
  if (ws_bytes < egs_splat_bwd_ws_bytes(n)) {
    set_error(EGS_ERR_WORKSPACE, "splat_bwd workspace too small", __FILE__, __LINE__);
    return EGS_ERR_WORKSPACE;
  }
  float* gpack = nullptr;
  int rc = splat_bwd_packed(n, patches, width, height, us, cinv2ds, alphas, colors, areas, pol, contrib, final_tau,
                            patch_range_per_tile, gsid_per_patch, dloss_dgammas, ws, ws_bytes, &gpack, stream, nullptr,
                            nullptr, nullptr);
  if (rc) return rc;
  return unpack_grads(n, gpack, dloss_dus, dloss_dcinv2ds, dloss_dalphas, dloss_dcolors, (hipStream_t)stream);
}
