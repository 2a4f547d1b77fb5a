/*
 * egs_hip.h -- C ABI of libegs_hip.so: the MI355X (gfx950) rasterizer hot path
 * that replaces the reference's CUDA extension `gsplatcu`.
 *
 * Drop-in boundary: the reference binds seven torch ops through
 * pybind11 (reference gsplatcu/ext.cpp:68-77, host wrappers gsplatcu/gausplat.cu).
 * Here the same seven ops are plain `extern "C"` functions over raw device
 * pointers + a HIP stream: no torch / pybind types cross this boundary.  The
 * Python mirror of the reference interface (easygaussiansplatting_amd/gsplatcu.py)
 * binds them with ctypes; INTEGRATION.md shows the stub a reference maintainer
 * would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in `_host`;
 *   - tensors are fp32 / int32 row-major contiguous, shapes as in SURVEY.md §8b;
 *   - `stream` is a hipStream_t (passed as void*); every call only ENQUEUES work
 *     on it -- no device-wide synchronisation, no allocation;
 *   - Jacobian pointers may be NULL (calc_J = False);
 *   - outputs need NO zero-fill by the caller: every op writes every row of every output it is
 *     handed, culled Gaussians as zeros -- what the reference's torch::full(.., 0) outputs read as
 *     (gausplat.cu:36-38,170,214,265,307,347) without the 528 B per Gaussian and step of fill
 *     kernels; per-Gaussian output rows leave the kernels as dwordx4 stores of whole 256-row spans,
 *     so output pointers of the five per-Gaussian ops and of egs_chain_rule must be 16-B aligned
 *     (checked; torch allocations are);
 *   - return value: 0 on success, otherwise a hipError_t (or EGS_ERR_*) and
 *     egs_last_error_string() describes it.
 */
#ifndef EGS_HIP_H_
#define EGS_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EGS_ABI_VERSION 9

#define EGS_ERR_BAD_ARG 10001
#define EGS_ERR_WORKSPACE 10002

/* Raster policy: which of the reference's three mutually inconsistent pipeline
 * definitions to follow.  One kernel family, runtime-uniform
 * parameters.  egs_policy_gsplatcu() == the CUDA extension (drop-in default),
 * egs_policy_forward_cpu() == forward_cpu.py / gsplat/gausplat.py. */
typedef struct EgsPolicy {
  int32_t near_cull;   /* project: z < 0.2 -> depth = -1, outputs stay 0   (kernel.cu:588)            */
  int32_t fov_mode;    /* 0: lim = 1.3*W/(2fx) (gausplat.cu:225) 1: 1.3*2*atan(W/(2fx)) (gausplat.py:136) 2: none */
  float det_eps;       /* added to det(cov2d)                              (gausplat.py:179)          */
  int32_t nan_cull;    /* inverseCov2D: NaN 1/det -> depth = -1            (kernel.cu:300-305)        */
  int32_t radius_mode; /* 0: ceil(3 sqrt|a|) (kernel.cu:308)  1: trunc(3 sqrt a) (gausplat.py:181)     */
  int32_t footprint;   /* 0: every pixel of every tile in the rect (kernel.cu:105-110)
                          1: the pixel box of gausplat.py:212-215                                      */
  int32_t far_cull;    /* splat: skip depth<0.2 || depth>100 || |u/[W,H]|>1.3 (gausplat.py:204,208)    */
  int32_t maha_floor;  /* max(0, m)                                        (kernel.cu:243)            */
  int32_t alpha_clamp; /* min(0.99, alpha')                                (kernel.cu:245)            */
  float alpha_skip;    /* skip alpha' < 0.002                              (kernel.cu:246)            */
  float tau_stop;      /* pixel done when tau < 1e-4                       (kernel.cu:256)            */
  int32_t depth_key;   /* 0: uint32(depth*1000) (kernel.cu:73)  1: raw fp32 bits (== argsort, gausplat.py:192) */
  int32_t nan_maha;    /* 0: a Gaussian whose conic or centre holds a NaN blends at min(0.99, alpha) into every pixel of
                          its tiles -- CUDA's max(0.0f, NaN) == 0 (kernel.cu:243-246, 909-913); 1: such pixels are
                          skipped (no NaN ever reaches the image).  Either way a NaN that arises at single pixels from
                          inf * 0 is skipped (oracle/gs_oracle.py NAN_MAHA)                                    */
} EgsPolicy;

void egs_policy_gsplatcu(EgsPolicy* p);
void egs_policy_forward_cpu(EgsPolicy* p);

int egs_abi_version(void);
const char* egs_last_error_string(void);

/* ---- per-Gaussian stages ------------------------------------------------ */

/* gsplatcu.project  (ext.cpp:54-61, gausplat.cu:253-296, kernel.cu:553-617).
 * pws[N,3], Rcw[9], tcw[3] -> us[N,2], pcs[N,3], depths[N], du_dpcs[N,2,3]|NULL */
int egs_project(int n, const float* pws, const float* Rcw, const float* tcw, float fx, float fy,
                float cx, float cy, const EgsPolicy* pol, float* us, float* pcs, float* depths,
                float* du_dpcs, void* stream);

/* gsplatcu.computeCov3D  (ext.cpp:39-42, gausplat.cu:162-199, kernel.cu:326-423).
 * rots[N,4] (w,x,y,z), scales[N,3], depths[N] -> cov3ds[N,6], dcov3d_drots[N,6,4], dcov3d_dscales[N,6,3] */
int egs_cov3d(int n, const float* rots, const float* scales, const float* depths, const EgsPolicy* pol,
              float* cov3ds, float* dcov3d_drots, float* dcov3d_dscales, void* stream);

/* gsplatcu.computeCov2D  (ext.cpp:44-52, gausplat.cu:201-251, kernel.cu:425-551).
 * -> cov2ds[N,3], dcov2d_dcov3ds[N,3,6], dcov2d_dpcs[N,3,3] */
int egs_cov2d(int n, const float* cov3ds, const float* pcs, const float* Rcw, const float* depths,
              float fx, float fy, float width, float height, const EgsPolicy* pol, float* cov2ds,
              float* dcov2d_dcov3ds, float* dcov2d_dpcs, void* stream);

/* gsplatcu.sh2Color  (ext.cpp:63-66, gausplat.cu:298-338, kernel.cu:619-807).
 * shs[N,K] (K in {3,12,27,48}), pws[N,3], twc[3] -> colors[N,3], dcolor_dshs[N,1,K/3], dcolor_dpws[N,3,3] */
int egs_sh2color(int n, int sh_dim, const float* shs, const float* pws, const float* twc,
                 float* colors, float* dcolor_dshs, float* dcolor_dpws, void* stream);

/* gsplatcu.inverseCov2D  (ext.cpp:34-36, gausplat.cu:340-373, kernel.cu:274-324).
 * cov2ds[N,3], depths[N] (IN/OUT: NaN -> -1) -> cinv2ds[N,3], areas[N,2] int32, dcinv2d_dcov2ds[N,3,3] */
int egs_inv_cov2d(int n, const float* cov2ds, float* depths, const EgsPolicy* pol, float* cinv2ds,
                  int32_t* areas, float* dcinv2d_dcov2ds, void* stream);

/* ---- splat = binning + sort + draw  (gausplat.cu:24-112) ------------------
 *
 * The reference needs P (number of tile-patches) on the host between the prefix
 * sum and the key expansion (gausplat.cu:67); so does a drop-in that returns
 * gsid_per_patch[P].  The op is therefore split in two enqueue-only calls with a
 * single 8-byte read-back (total_patches[0..1]) between them (done by the caller on its own stream):
 *
 *   egs_splat_bin()   getRects (kernel.cu:82-122) + depth-key sort of the N
 *                     Gaussians + exclusive scan  -> *total_patches (device u32)
 *   egs_splat_draw()  createKeys (kernel.cu:46-80) in depth order + stable tile
 *                     sort + getRanges (kernel.cu:125-150) + draw (kernel.cu:152-271)
 *
 * Sorting the N Gaussians by depth key first and the P patches by tile id second
 * (both stable LSD radix) yields exactly the order of the reference's single
 * sort on (tile<<32 | depth_key): ties in index order.
 */
size_t egs_splat_bin_ws_bytes(int n);
size_t egs_splat_draw_ws_bytes(int n, int64_t patches, int width, int height);

/* us[N,2], areas[N,2] (IN/OUT), depths[N] (IN/OUT) ; ws_bin must stay alive until
 * egs_splat_draw() has been enqueued.
 * total_patches: device uint32[2] -- [0] = P, [1] = the largest depth key of this call.
 * key_bits_hint: number of low depth-key bits to sort (0 or 32 = all).  A caller that passes a
 *   smaller hint (e.g. bit_length of the previous call's max key + 1) MUST check
 *   total_patches[1] < 2^hint after the read-back and, if not, call egs_splat_bin again with
 *   hint 32 (the call is idempotent). */
int egs_splat_bin(int n, int width, int height, const float* us, int32_t* areas, float* depths,
                  const EgsPolicy* pol, int key_bits_hint, void* ws_bin, size_t ws_bin_bytes,
                  uint32_t* total_patches, void* stream);

/* patches = the value read back from total_patches.  Outputs, all fully written here (no
 * zero-fill needed): image[3,H,W], contrib[H,W] int32, final_tau[H,W] (empty tiles get
 * image = 0, contrib = 0, final_tau = 0 like the reference), patch_range_per_tile[T,2] int32,
 * gsid_per_patch[P] int32. */
int egs_splat_draw(int n, int64_t patches, int width, int height, const float* us,
                   const float* cinv2ds, const float* alphas, const float* colors, const int32_t* areas,
                   const EgsPolicy* pol, const void* ws_bin, void* ws_draw, size_t ws_draw_bytes,
                   float* image, int32_t* contrib, float* final_tau, int32_t* patch_range_per_tile,
                   int32_t* gsid_per_patch, void* stream);

/* The same two stages for a host that does not want the GPU to wait for its read of total_patches (the reference
 * idles around cudaMemcpy(&P), gausplat.cu:67): egs_splat_bin_mb also stores {P, max depth key} into host_totals
 * (a page-locked mailbox slot, egs_mailbox_slot / _arm / _fetch below), and egs_splat_draw_dev is enqueued right
 * behind it with buffers sized by patch_capacity (gsid_per_patch, egs_splat_draw_ws_bytes(n, patch_capacity, ..))
 * and the real count taken from total_patches[0] on the device.  The host then reads the slot: if the count
 * exceeds patch_capacity (nothing was written out of bounds) or the depth keys outgrew the hint, the stages are
 * redone the synchronous way.  gsplatcu.splat works like this from the second call of a problem size on. */
int egs_splat_bin_mb(int n, int width, int height, const float* us, int32_t* areas, float* depths,
                     const EgsPolicy* pol, int key_bits_hint, void* ws_bin, size_t ws_bin_bytes,
                     uint32_t* total_patches, uint32_t* host_totals /*nullable*/, void* stream);
int egs_splat_draw_dev(int n, int64_t patch_capacity, const uint32_t* total_patches, int width, int height,
                       const float* us, const float* cinv2ds, const float* alphas, const float* colors,
                       const int32_t* areas, const EgsPolicy* pol, const void* ws_bin, void* ws_draw,
                       size_t ws_draw_bytes, float* image, int32_t* contrib, float* final_tau,
                       int32_t* patch_range_per_tile, int32_t* gsid_per_patch, void* stream);

/* The packed 48-byte 2D records the draw kernels gather (one aligned record per list entry instead of the
 * reference's four gathers, fetch2shared kernel.cu:13-44) as a CALLER-HELD buffer rec[N][12]: gsplatcu.splat packs
 * once, draws with egs_splat_draw_rec / _rec_dev and keeps the buffer for the splatB that follows with the same
 * tensors (egs_splat_bwd_rec; tile_order nullable = the [order | work] buffer of that draw). */
int egs_pack_records(int n, int width, int height, const float* us, const float* cinv2ds, const float* alphas,
                     const float* colors, const int32_t* areas /*pixel-box policy only*/, const EgsPolicy* pol,
                     void* rec, void* stream);
/* splat (ext.cpp:10-18, gausplat.cu:24-112), tile-footprint policies with a skip threshold (pol->footprint == 0,
 * pol->alpha_skip > 0): egs_pack_records and egs_splat_bin(_mb) as ONE pass over the 2D Gaussians.  `rec` receives the
 * packed records; the binning state left in ws_bin makes egs_splat_draw_rec* (flags = EGS_DRAW_MASKED_LISTS) emit the
 * reference's lists with exact block masks in their values.  depths / areas are updated in place as by egs_splat_bin
 * (kernel.cu:114-119).  host_totals nullable (page-locked mailbox slot, see egs_splat_bin_mb). */
int egs_splat_bin_pack(int n, int width, int height, const float* us, const float* cinv2ds, const float* alphas,
                       const float* colors, int32_t* areas, float* depths, const EgsPolicy* pol, int key_bits_hint,
                       void* ws_bin, size_t ws_bin_bytes, uint32_t* total_patches, uint32_t* host_totals, void* rec,
                       uint32_t* stamp /* nullable: egs_pair_stamp_words(n) words, content stamps */,
                       uint8_t* visible /* nullable: n bytes, depths > 0.2 after the in-place cull (gsmodel.py:50) */,
                       void* stream);
/* egs_splat_draw_rec / egs_splat_draw_rec_dev for the seven-op surface (flags = EGS_DRAW_MASKED_LISTS): gsid_per_patch
 * receives the list the draw kernels walk (with masks), gsid_plain (nullable) the list splat's caller gets
 * (gausplat.cu:108-111), written by the range kernel on its way: no egs_strip_list_masks launch. */
int egs_splat_draw_rec_plain(int n, int64_t patches, int width, int height, const void* rec, const EgsPolicy* pol,
                             const void* ws_bin, void* ws_draw, size_t ws_draw_bytes, float* image, int32_t* contrib,
                             float* final_tau, int32_t* patch_range_per_tile, int32_t* gsid_per_patch,
                             int32_t* gsid_plain, int32_t* tile_order /*nullable*/, float* grad_records /*nullable*/,
                             int flags, void* stream);
int egs_splat_draw_rec_dev_plain(int n, int64_t patch_capacity, const uint32_t* total_patches, int width, int height,
                                 const void* rec, const EgsPolicy* pol, const void* ws_bin, void* ws_draw,
                                 size_t ws_draw_bytes, float* image, int32_t* contrib, float* final_tau,
                                 int32_t* patch_range_per_tile, int32_t* gsid_per_patch, int32_t* gsid_plain,
                                 int32_t* tile_order /*nullable*/, float* grad_records /*nullable*/, int flags,
                                 void* stream);
/* Content-validated pairing of the reference's two independent calls splat and splatB.  splat keeps the list with
 * masks and a STAMP of the us / cinv2ds / alphas values it was built from (two position-dependent 32-bit sums per 256
 * Gaussians, written by egs_splat_bin_pack).  splatB calls egs_pack_records_validate: it packs the records from the
 * tensors splatB was given (always fresh), stamps them (stamp_b), and REPAIRS the kept list on the device: an entry
 * that is not the caller's own (plain[i] differs) or whose Gaussian lies in a block of 256 with a different stamp
 * becomes the caller's entry with all four blocks set (valid for any data).  egs_splat_bwd_rec_lists(.., kept, flags =
 * EGS_DRAW_MASKED_LISTS) then walks a list that IS the caller's: no pointer or version comparison is involved, a write
 * through tensor.data or another library's kernel between the two calls is seen. */
size_t egs_pair_stamp_words(int n);
int egs_pack_records_validate(int n, int width, int height, const float* us, const float* cinv2ds,
                              const float* alphas, const float* colors, const EgsPolicy* pol, void* rec,
                              const uint32_t* stamp_a, uint32_t* stamp_b, int64_t patches, void* kept,
                              const int32_t* plain, void* stream);
/* *flag = 1 if any of the n_words 32-bit words of the device buffers a and b differs (never cleared: the caller zeroes
 * it and may run several comparisons into one flag); 4-byte aligned pointers, 16-byte aligned ones are compared sixteen
 * bytes per lane.  The host layer's content check of the public splat / splatB pair (ext.cpp:10-32 are two independent
 * calls; what splat keeps for splatB is used only if splatB is handed the same VALUES): ABI 9. */
int egs_words_differ(const void* a, const void* b, int64_t n_words, int32_t* flag, void* stream);
/* plain[i] = masked[i] & 0x0FFFFFFF for i < min(count, *count_dev) (count_dev nullable: a device-side patch count
 * the host has not read yet): gsid_per_patch as the reference returns it (gausplat.cu:108-111). */
int egs_strip_list_masks(int64_t count, const uint32_t* count_dev, const void* masked, int32_t* plain, void* stream);
/* egs_splat_bwd_rec with flags = EGS_DRAW_MASKED_LISTS: gsid_per_patch is the masked list the forward draw walked. */
int egs_splat_bwd_rec_lists(int n, int64_t patches, int width, int height, const void* rec, const EgsPolicy* pol,
                            const int32_t* contrib, const float* final_tau, const int32_t* patch_range_per_tile,
                            const int32_t* gsid_per_patch, const float* dloss_dgammas, void* ws, size_t ws_bytes,
                            const int32_t* tile_order, float* grad_records, float* dloss_dus, float* dloss_dcinv2ds,
                            float* dloss_dalphas, float* dloss_dcolors, int flags, void* stream);
int egs_splat_bwd_rec(int n, int64_t patches, int width, int height, const void* rec, const EgsPolicy* pol,
                      const int32_t* contrib, const float* final_tau, const int32_t* patch_range_per_tile,
                      const int32_t* gsid_per_patch, const float* dloss_dgammas, void* ws, size_t ws_bytes,
                      const int32_t* tile_order /*nullable*/, float* grad_records /*nullable: zeroed by that draw*/,
                      float* dloss_dus, float* dloss_dcinv2ds, float* dloss_dalphas, float* dloss_dcolors,
                      void* stream);

/* gsplatcu.splatB  (ext.cpp:20-32, gausplat.cu:114-159, kernel.cu:809-950).
 * Gradient outputs (fully written): dloss_dus[N,2], dloss_dcinv2ds[N,3],
 * dloss_dalphas[N], dloss_dcolors[N,3].  ws: egs_splat_bwd_ws_bytes(n). */
size_t egs_splat_bwd_ws_bytes(int n);
int egs_splat_bwd(int n, int64_t patches, int width, int height, const float* us, const float* cinv2ds,
                  const float* alphas, const float* colors, const int32_t* areas, const EgsPolicy* pol,
                  const int32_t* contrib, const float* final_tau, const int32_t* patch_range_per_tile,
                  const int32_t* gsid_per_patch, const float* dloss_dgammas, void* ws, size_t ws_bytes,
                  float* dloss_dus, float* dloss_dcinv2ds, float* dloss_dalphas, float* dloss_dcolors,
                  void* stream);

/* ---- building blocks, exported for the parity tests ---------------------- */

/* Stable LSD radix sort of (uint32 key, uint32 value) pairs on key bits
 * [begin_bit, end_bit).  Replaces thrust::sort_by_key (gausplat.cu:82).
 * keys/vals are ping-ponged with keys_alt/vals_alt; *result_in_alt_host tells
 * where the result is (decided on the host from the pass count). */
size_t egs_sort_pairs_ws_bytes(int64_t n);
int egs_sort_pairs(int64_t n, uint32_t* keys, uint32_t* vals, uint32_t* keys_alt, uint32_t* vals_alt,
                   int begin_bit, int end_bit, void* ws, size_t ws_bytes, int* result_in_alt_host,
                   void* stream);

/* Exclusive prefix sum of uint32 (replaces thrust::inclusive_scan, gausplat.cu:64);
 * out[i] = sum_{j<i} in[gather ? gather[j] : j]; *total (device) = sum of all. */
size_t egs_scan_ws_bytes(int64_t n);
int egs_exclusive_scan_u32(int64_t n, const uint32_t* in, const uint32_t* gather, uint32_t* out,
                           uint32_t* total, void* ws, size_t ws_bytes, void* stream);

/* ---- fused chain rule --------------------
 * gsmodel.py:71-85 == backward_cpu.py:476-482 in one pass over the Jacobians. */
int egs_chain_rule(int n, int sh_dim, const float* dloss_dus, const float* dloss_dcinv2ds,
                   const float* dloss_dcolors, const float* Rcw, const float* dcinv2d_dcov2ds,
                   const float* dcov2d_dcov3ds, const float* dcov3d_drots, const float* dcov3d_dscales,
                   const float* dcolor_dshs, const float* du_dpcs, const float* dcov2d_dpcs,
                   const float* dcolor_dpws, float* dloss_dpws, float* dloss_dshs, float* dloss_dscales,
                   float* dloss_drots, void* stream);

/* ---- fused training path -------------------------------------
 * What GSFunction.forward / .backward (gsplat/gsmodel.py:6-93) need, without the
 * 436 B/Gaussian of Jacobians crossing HBM:
 *   egs_fused_forward   = project + computeCov3D + computeCov2D + sh2Color + inverseCov2D in ONE
 *                         kernel, followed by egs_splat_bin (then read *total_patches and call
 *                         egs_splat_draw as for `splat`);
 *   egs_fused_backward  = splatB's draw pass into packed gradient records + ONE kernel that
 *                         re-derives the Jacobians in registers and applies backward.md
 *                         eq (3)(4)(5)(7) (gsmodel.py:71-85).
 * `depths`/`areas` are the arrays egs_fused_forward produced (incl. the in-place culling). */
/* rec (nullable): 48 N bytes; receives the packed 2D records of the draw kernels so that
 * egs_splat_draw_rec / egs_fused_backward skip their own packing pass.  With rec given, each of
 * us / cinv2ds / colors / areas may be NULL (they are only needed to continue on the seven-op surface)
 * and egs_fused_backward accepts NULL for them too.
 * visible (nullable): N bytes; receives depths[i] > 0.2 AFTER the in-place culling of splat, i.e. the
 * mask GSFunction.forward returns (gsmodel.py:50).
 * host_totals (nullable): device-visible address of a page-locked host uint32[2] (egs_mailbox_slot) that the
 * binning kernels write {P, max depth key} into as well -- the enqueue-ahead path then needs no copy. */
int egs_fused_forward(int n, int sh_dim, const float* pws, const float* rots, const float* scales,
                      const float* shs, const float* alphas, const float* Rcw, const float* tcw,
                      const float* twc, float fx, float fy, float cx, float cy, int width, int height,
                      const EgsPolicy* pol, float* us, float* depths, float* cinv2ds, float* colors,
                      int32_t* areas, void* rec, uint8_t* visible, float* dcolor_dpws /*nullable*/, int cull_lists,
                      int key_bits_hint, void* ws_bin, size_t ws_bin_bytes, uint32_t* total_patches,
                      uint32_t* host_totals, void* stream);
/* dcolor_dpws (nullable, [N][9] floats, 16-B aligned): dcolor/dpw of every Gaussian (what sh2Color's calc_J hands
 * back as dcolor_dpws, gausplat.cu:298-338), kept for the backward pass: egs_fused_backward given the same pointer
 * never reads the SH coefficients again -- eq (7)'s colour term is the only thing it needs them for, dL/dsh needs
 * the basis alone -- i.e. 36 B written here for 4 sh_dim bytes (192 at SH degree 3) not re-read there. */
int egs_splat_draw_rec(int n, int64_t patches, int width, int height, const void* rec, const EgsPolicy* pol,
                       const void* ws_bin, void* ws_draw, size_t ws_draw_bytes, float* image,
                       int32_t* contrib, float* final_tau, int32_t* patch_range_per_tile,
                       int32_t* gsid_per_patch, int32_t* tile_order /*nullable*/,
                       float* grad_records /*nullable*/, const int32_t* prev_tile_work /*nullable*/,
                       int order_ready, int flags /* EGS_DRAW_CULLED_LISTS or 0 */, void* stream);
/* prev_tile_work (nullable, T ints): the work part of the tile_order buffer an EARLIER render of the same camera
 * left behind; the forward dispatch order then sorts by it instead of by the list lengths.  Any values are
 * legal (every permutation of the tiles gives the same image), good ones balance the launch.  It may be the work
 * part of tile_order itself (a caller that keeps ONE buffer per camera: the order is refreshed in place).
 * order_ready != 0: tile_order already holds a dispatch order from an earlier render through the same buffer;
 * the draw uses it as it stands (no k_tile_order launch, 10 us) and only rewrites the work part.  The work
 * pattern of a camera drifts slowly, so a caller refreshes every few renders, not every time. */
/* grad_records (nullable, [N][12] floats): the packed per-Gaussian gradient records of the COMING backward pass;
 * the draw kernel zeroes them on the side (it is VALU-bound, the memory system idles) and egs_fused_backward,
 * given the same pointer, skips its own 48 N-byte fill.  Valid for ONE backward pass.
 * The draw kernels hand the tiles to the SIMDs longest list first (k_tile_order, one workgroup, after the
 * tile ranges are known).  tile_order (nullable, egs_tile_order_len(width, height) ints) receives that dispatch
 * order so that egs_fused_backward can reuse it instead of computing its own. */
size_t egs_tile_order_len(int width, int height);   /* ints: [forward dispatch order | per-tile work measured by the
                                                      * draw | per-tile walk length (segment path only)] */
/* Long lists split over several waves (reference: 256 threads per tile, kernel.cu:152-271 launched (16, 16) at
 * gausplat.cu:94; here a tile is ONE wave64 whose run time is the length of its walk, so a heavy-tailed scene -- a few
 * tiles with 10 000 entries, nothing saturating after reset_alpha, gsmodel.py:320-324 -- ends when its longest tile
 * ends).  egs_splat_draw_rec_seg is egs_splat_draw_rec (total_patches == NULL: `patches` exact) / egs_splat_draw_rec_dev
 * (total_patches on the device, `patches` the capacity) with a workspace of egs_seg_ws_bytes(patch capacity, ..) bytes:
 * tiles of more than `split_min` entries are walked in segments of `segment_len` entries -- front-to-back blending is
 * associative on (colour, tau) pairs -- by one wave each where this camera's previous render predicts the walk
 * (flags & EGS_DRAW_SEG_HISTORY: the walk part of tile_order is that render's), sequentially otherwise; either way the
 * workspace then holds, per segment end and pixel, the transmittance and the colour of everything behind it, and
 * egs_fused_backward(_raw) given the SAME workspace walks every segment with a wave of its own (no sequential
 * dependence is left in the backward pass).  Tile-footprint policies with alpha_skip > 0 and tau_stop > 0 only.
 * Images / contrib / final_tau equal the unsplit kernels' up to the rounding of  sum_s T_s C_s  against one running sum.
 * seg_hint (nullable, page-locked host memory, e.g. a mailbox slot, two words): [0] receives the longest list, [1] the
 * longest WALK of a recent render.  ABI 8: walk_word (nullable) is a PERSISTENT device word of the caller's, one per
 * problem size and stream, holding -1 before its first use: the draw items of this call gather the render's longest walk
 * in it, and the range kernel at the start of the NEXT call's draw stage on the stream publishes it into seg_hint[1] and
 * resets it -- the host word only ever holds the maximum of a completed render, from every render, split or not -- a
 * host that finds it below split_min may drop the workspace for later renders of the scene (the unsplit kernels are
 * then the same work with two launches less).  prev_tile_work / order_ready are ignored when the lists are split
 * (the work items are re-planned per render, by k_seg_plan). */
#define EGS_DRAW_SEG_HISTORY 4
/* flags of egs_splat_draw_rec_seg: take a tile's list length as its predicted walk (all its segments get a wave at once)
 * -- for a camera WITHOUT a walk on record, and (ABI 8) for tiles whose recorded walk is less than a quarter of their
 * list: a record from before reset_alpha.  For scenes the host knows to be walked to (nearly) their ends -- the
 * longest walk a recent render of the scene reported is a good part of its longest list, as right after reset_alpha --
 * where the alternative (one wave continuing from segment 1) is the serial tail this path exists to remove.  On a
 * saturating scene it would blend segments nobody looks at: exact either way, the balance is the host's call. */
#define EGS_DRAW_SEG_SPECULATE 8
size_t egs_seg_ws_bytes(int64_t patch_capacity, int width, int height);
/* segment_len (a power of two >= 64) / split_min: 0 keeps the current value; out2 (nullable) receives the values BEFORE the
 * call.  Process-wide default (256 / 1024, or EGS_SEG_L / EGS_SEG_MIN from the environment), one atomic word: a render
 * reads it ONCE, when its forward pass plans; the plan leaves L in the workspace header and every later launch -- the
 * backward pass included -- takes it from there and sizes its grid by the workspace (ABI 8: a change between a forward
 * and its backward call no longer matters).  A workspace sized under a larger setting than the one planned with simply
 * has fewer slots than tiles could use: tiles that do not fit stay unsplit. */
int egs_seg_config(int segment_len, int split_min, int* out2);
int egs_splat_draw_rec_seg(int n, int64_t patches, const uint32_t* total_patches /*nullable*/, int width, int height,
                           const void* rec, const EgsPolicy* pol, const void* ws_bin, void* ws_draw,
                           size_t ws_draw_bytes, float* image, int32_t* contrib, float* final_tau,
                           int32_t* patch_range_per_tile, int32_t* gsid_per_patch, int32_t* tile_order /*nullable*/,
                           float* grad_records /*nullable*/, const int32_t* prev_tile_work /*nullable*/, int order_ready,
                           int flags, void* seg_ws /*nullable: the unsplit draw stage*/, size_t seg_ws_bytes,
                           uint32_t* seg_hint /*nullable*/, int32_t* walk_word /*nullable*/,
                           int32_t* gsid_plain /*nullable; with EGS_DRAW_MASKED_LISTS: the list without its masks, what
                                                 the caller of `splat` gets (as egs_splat_draw_rec_plain)*/,
                           void* stream);
/* splatB for a host that may or may not have kept what its forward pass left (the seven-op surface: `splatB` is handed
 * tensors).  rec (nullable: packed here from us / cinv2ds / alphas / colors), tile_order / grad_records (nullable: the
 * forward draw's [order | work | walk] buffer and cleared gradient records), flags as egs_splat_bwd_rec_lists.
 * seg_ws == NULL: the unsplit kernel (egs_splat_bwd / egs_splat_bwd_rec_lists).  seg_ws + rebuild == 0: the workspace the
 * forward's egs_splat_draw_rec_seg filled.  seg_ws of egs_seg_rebuild_ws_bytes(patches, ..) + rebuild != 0: nothing
 * was kept -- every tile's walk is read off `contrib`, the forward segment launches run once more over [0, walk) with
 * their pixels going to scratch (they only rebuild the segment-end states: about the cost of a forward draw), then
 * every segment is walked backward by a wave of its own: on scene.skewed_scene after reset_alpha 3.4 ms of one-wave-
 * per-tile backward draw become ~1.2 ms.  seg_hint (nullable): page-locked words that learn the longest walk from
 * either path -- a host decides from them whether its next call brings a workspace. */
size_t egs_seg_rebuild_ws_bytes(int64_t patch_capacity, int width, int height);
int egs_splat_bwd_seg(int n, int64_t patches, int width, int height, const float* us, const float* cinv2ds,
                      const float* alphas, const float* colors, const void* rec /*nullable*/, const EgsPolicy* pol,
                      const int32_t* contrib, const float* final_tau, const int32_t* patch_range_per_tile,
                      const int32_t* gsid_per_patch, const float* dloss_dgammas, void* ws, size_t ws_bytes,
                      const int32_t* tile_order /*nullable*/, float* grad_records /*nullable*/, float* dloss_dus,
                      float* dloss_dcinv2ds, float* dloss_dalphas, float* dloss_dcolors, int flags,
                      void* seg_ws /*nullable*/, size_t seg_ws_bytes, int rebuild, uint32_t* seg_hint /*nullable*/,
                      void* stream);
/* As egs_splat_draw_rec, for a host that enqueues the draw stage BEFORE it has read total_patches (no GPU
 * idle time around the read-back): patch_capacity sizes gsid_per_patch and ws_draw
 * (egs_splat_draw_ws_bytes(n, patch_capacity, ..)), the real patch count is taken from total_patches[0] on
 * the device.  host_totals (nullable, page-locked host uint32[2]) receives total_patches[0..1] by an
 * asynchronous copy enqueued in front of the draw stage.  The caller checks afterwards: if the count exceeds
 * patch_capacity the outputs are incomplete (nothing is written out of bounds) and the stage must be redone
 * with the exact count (egs_splat_draw_rec). */
int egs_splat_draw_rec_dev(int n, int64_t patch_capacity, const uint32_t* total_patches, uint32_t* host_totals,
                           int width, int height, const void* rec, const EgsPolicy* pol, const void* ws_bin,
                           void* ws_draw, size_t ws_draw_bytes, float* image, int32_t* contrib, float* final_tau,
                           int32_t* patch_range_per_tile, int32_t* gsid_per_patch, int32_t* tile_order /*nullable*/,
                           float* grad_records /*nullable*/, const int32_t* prev_tile_work /*nullable*/,
                           int order_ready, int flags, void* stream);
/* Measurement helper (bench.py): one device-to-device float4 copy of `bytes` (multiple of 16, both pointers
 * 16-B aligned) -- the achievable-HBM-bandwidth probe SURVEY 8(d) asks the roofline to be quoted against. */
int egs_hbm_copy_probe(void* dst, const void* src, size_t bytes, void* stream);
/* Measurement helper (bench.py): the shader clock under full VALU load -- a chain of dependent v_fma on every SIMD,
 * bracketed by the shader-clock counter (s_memtime) and the constant 100-MHz counter (s_memrealtime).  out: 8 uint64 in
 * device memory; once the stream has run it, MHz = (out[1] - out[0]) / (out[3] - out[2]) * 100.  What a VALU-bound
 * kernel's roofline is priced against: 1024 SIMDs x this clock. */
int egs_clock_probe(void* out8, int iters, void* stream);
/* Mailbox for that read-back: `slots` page-locked landing zones, each with a HIP event.  egs_mailbox_post
 * enqueues the asynchronous 8-byte copy of total_patches[0..1] into a slot on `stream` and records the
 * slot's event behind it; egs_mailbox_fetch returns 1 and the two words once the copy has landed, 0 when it
 * has not (blocking == 0), or waits for it on the slot's event (blocking != 0): ONE C-side wait where the
 * reference blocks in cudaMemcpy (gausplat.cu:67); a negative value is -(error code).  A slot may be re-posted
 * after it was fetched. */
void* egs_mailbox_create(int slots);
void egs_mailbox_destroy(void* mailbox);
int egs_mailbox_post(void* mailbox, int slot, const uint32_t* total_patches, void* stream);
/* The same without the copy and without an event: egs_mailbox_slot is the slot's address (valid on host and
 * device: pass it as host_totals of egs_fused_forward, whose kernels store the two words there);
 * egs_mailbox_arm -- called BEFORE that enqueue -- marks the slot empty, and egs_mailbox_fetch then polls the
 * slot's first word (`stream` is only used to recover if nothing ever arrives). */
uint32_t* egs_mailbox_slot(void* mailbox, int slot);
int egs_mailbox_arm(void* mailbox, int slot, void* stream);
int egs_mailbox_fetch(void* mailbox, int slot, int blocking, uint32_t* out2);
/* the four words of a slot as they stand right now (no waiting, no state change): for slots used as a landing zone of
 * hints (egs_splat_draw_rec_seg's seg_hint); egs_mailbox_clear sets them to 0xFFFFFFFF ("nothing yet"). */
int egs_mailbox_peek(void* mailbox, int slot, uint32_t* out4);
int egs_mailbox_clear(void* mailbox, int slot);
size_t egs_fused_backward_ws_bytes(int n);
/* OR-ed into `phase`: the forward pass that filled tile_order was itself dispatched by measured work
 * (prev_tile_work != NULL), so the backward pass keeps that order instead of sorting the tiles again by the
 * work this render measured (one k_tile_order launch, 8 us, less; without the flag it sorts). */
#define EGS_BWD_KEEP_FORWARD_ORDER 16
/* OR-ed into `phase` / passed as `flags` of egs_splat_draw_rec*: the lists are the FOOTPRINT-CULLED ones of
 * egs_fused_forward(cull_lists = 1).  A Gaussian is then listed only for the tiles of its rect (getRects,
 * kernel.cu:82-122) that its footprint {alpha' >= alpha_skip} can reach -- the tiles left out are tiles in which the
 * reference `continue`s on every pixel (kernel.cu:246), so images and gradients are unchanged, but list positions
 * (and with them `contrib`) refer to the culled lists -- and every list value carries, above the low 28 bits of the
 * Gaussian index, the 4-bit mask of the 8x8 pixel blocks of the tile the footprint reaches.  Internal to the fused
 * path: the seven-op surface (egs_splat_bin / egs_splat_draw) always produces the reference's lists. */
#define EGS_BWD_CULLED_LISTS 32
/* OR-ed into `phase` of egs_fused_backward(_raw): the parameter-gradient outputs (dloss_dpws, dloss_dshs | low + high,
 * dloss_dalphas, dloss_dscales, dloss_drots) already hold the gradients of earlier views of the step; this view's are
 * ADDED to them by the chain-rule kernel (dloss_dus is per view and always written).  Replaces autograd's separate
 * accumulation kernels for a rank that renders several views per step (bench.py --views-per-rank, Trainer.step). */
#define EGS_BWD_ACCUMULATE 64
/* OR-ed into `phase` of egs_fused_backward(_raw): the SH gradient of this view stays in its FACTORED form.  Eq (5)
 * (gsmodel.py:84-85: dL/dshs = dL/dcolors @ dcolor/dshs) is an outer product per Gaussian -- dL/dcolour[rgb] times the
 * SH basis of the direction from the camera centre -- so `dloss_dshs` (raw: `dloss_dlow_shs`) receives the THREE
 * floats dL/dcolour per Gaussian ([N][3]; always written, never accumulated; zero for a Gaussian this view did not
 * draw) FOLLOWED by the view's camera centre twc[3] -- 3 n + 3 floats, one row of egs_sh_grad_views' input -- and
 * `dloss_dhigh_shs` is not touched (may be NULL).  The rows are formed once per step, for all views, by
 * egs_sh_grad_views.  A host that renders V views per step writes 12 instead of 4 sh_dim bytes per Gaussian and view;
 * a data-parallel host all-gathers 12 bytes per Gaussian and VIEW instead of all-reducing 4 sh_dim per Gaussian
 * (192 of the 236 bytes of SURVEY 8e's exchange at degree 3). */
#define EGS_BWD_FACTORED_SH 128
#define EGS_DRAW_CULLED_LISTS 1
/* flags of egs_splat_draw_rec* / egs_splat_bwd_rec_lists: the lists are the REFERENCE's complete lists (every tile of
 * every rect, kernel.cu:46-80) whose values carry the same 4-bit block masks above the Gaussian index -- what
 * egs_splat_bin_pack prepares.  The draw kernels take the masks instead of testing the record's certain-miss box per
 * entry (1.95 instead of 2.35 block evaluations per entry, entries with an empty mask cost a list read); images and
 * gradients are unchanged (the pixels skipped are pixels the reference `continue`s on, kernel.cu:246).  The caller
 * hands gsid_per_patch back to ITS caller without the masks (egs_splat_draw_rec_plain, or egs_strip_list_masks). */
#define EGS_DRAW_MASKED_LISTS 2
/* phase 0: the whole backward pass.  phase 1: only splatB's draw pass (packed gradient records -> ws).
 * phase 2: only the per-Gaussian chain rule for rows [row_begin, row_begin + row_count), row_begin a multiple
 * of 256, reading the records phase 1 left in the SAME ws: a data-parallel caller launches the rows in a few
 * chunks and hands each chunk's gradients to RCCL while the next chunk is computed (dist_views.ChunkedExchange). */
int egs_fused_backward(int n, int sh_dim, int64_t patches, int width, int height, const float* pws,
                       const float* rots, const float* scales, const float* shs, const float* alphas,
                       const float* Rcw, const float* tcw, const float* twc, float fx, float fy, float cx,
                       float cy, const EgsPolicy* pol, const float* us, const float* cinv2ds,
                       const float* colors, const int32_t* areas, const void* rec /*nullable*/,
                       const float* depths,
                       const int32_t* contrib, const float* final_tau, const int32_t* patch_range_per_tile,
                       const int32_t* gsid_per_patch, const float* dloss_dgammas, void* ws, size_t ws_bytes,
                       float* dloss_dpws, float* dloss_dshs, float* dloss_dalphas, float* dloss_dscales,
                       float* dloss_drots, float* dloss_dus, const int32_t* tile_order /*nullable*/,
                       float* grad_records /*nullable: zeroed by the forward draw*/,
                       const float* dcolor_dpws /*nullable: left by egs_fused_forward*/, int phase, int row_begin,
                       int row_count, void* seg_ws /*nullable: the forward's egs_splat_draw_rec_seg workspace*/,
                       size_t seg_ws_bytes, void* stream);

/* The same pair on the OPTIMIZER's tensors (gsplat/gsmodel.py:96-129: alphas_raw, scales_raw, rots_raw,
 * low_shs [N,3], high_shs [N,sh_dim-3]): the activations of gsplat/utils.py:121-150 (sigmoid, exp,
 * normalize, cat) are applied inside the kernels and the gradients come back with respect to the raw
 * tensors, i.e. GSModel.forward (gsmodel.py:185-212) + GSFunction in two calls.  `rec` is required (the
 * activated alpha only exists inside the records). */
int egs_fused_forward_raw(int n, int sh_dim, const float* pws, const float* rots_raw, const float* scales_raw,
                          const float* low_shs, const float* high_shs, const float* alphas_raw, const float* Rcw,
                          const float* tcw, const float* twc, float fx, float fy, float cx, float cy, int width,
                          int height, const EgsPolicy* pol, float* us, float* depths, float* cinv2ds, float* colors,
                          int32_t* areas, void* rec, uint8_t* visible, float* dcolor_dpws /*nullable*/,
                          int cull_lists, int key_bits_hint, void* ws_bin, size_t ws_bin_bytes,
                          uint32_t* total_patches, uint32_t* host_totals, void* stream);
int egs_fused_backward_raw(int n, int sh_dim, int64_t patches, int width, int height, const float* pws,
                           const float* rots_raw, const float* scales_raw, const float* low_shs,
                           const float* high_shs, const float* alphas_raw, const float* Rcw, const float* tcw,
                           const float* twc, float fx, float fy, float cx, float cy, const EgsPolicy* pol,
                           const float* us, const float* cinv2ds, const float* colors, const int32_t* areas,
                           const void* rec, const float* depths, const int32_t* contrib, const float* final_tau,
                           const int32_t* patch_range_per_tile, const int32_t* gsid_per_patch,
                           const float* dloss_dgammas, void* ws, size_t ws_bytes, float* dloss_dpws,
                           float* dloss_dlow_shs, float* dloss_dhigh_shs, float* dloss_dalphas_raw,
                           float* dloss_dscales_raw, float* dloss_drots_raw, float* dloss_dus,
                           const int32_t* tile_order /*nullable*/, float* grad_records /*nullable*/,
                           const float* dcolor_dpws /*nullable*/, int phase, int row_begin, int row_count,
                           void* seg_ws /*nullable*/, size_t seg_ws_bytes, void* stream);
/* The SH-coefficient gradient of a step from the factored form EGS_BWD_FACTORED_SH leaves:
 *     dloss_dshs[i][c][rgb] (+)= scale * sum_v  rows[v][3 i + rgb] * basis_c(pws[i] - twc_v)
 * rows: `views` rows of `row_stride` floats, row v = { dL/dcolour of view v [N][3], twc_v[3], padding } -- this rank's
 * views or the all-gathered rows of every rank (then scale = 1 / ranks for the mean the exchange of SURVEY 8e forms).
 * dloss_dhigh_shs == NULL: dloss_dshs is [N][sh_dim]; else the raw layout (dloss_dshs = low [N][3], high
 * [N][sh_dim - 3]).  accumulate != 0: added to what the outputs hold.  (Replaces, per step instead of per view, the
 * dL/dshs rows of gsmodel.py:84-85.) */
int egs_sh_grad_views(int n, int sh_dim, int views, const float* pws, const float* rows, int64_t row_stride,
                      float scale, float* dloss_dshs, float* dloss_dhigh_shs /*nullable*/, int accumulate,
                      void* stream);


/* ---- fused training loss -----------------------------------------
 * gau_loss = (1 - lambda) * mean|image - gt| + lambda * (1 - SSIM(image, gt)), SSIM with the
 * 11x11 Gaussian window (sigma 1.5, zero padding) of reference gsplat/pytorch_ssim.py:10-66.
 * image, gt_image: [3,H,W].  loss_out: device float[3] = {loss, l1, ssim}.  dloss_dimage
 * (nullable) [3,H,W] receives grad_scale * dloss/dimage -- exactly the dloss_dgammas that
 * splatB / egs_fused_backward consume. */
size_t egs_gau_loss_ws_bytes(int height, int width);
int egs_gau_loss(int height, int width, const float* image, const float* gt_image, float loss_lambda,
                 float grad_scale, void* ws, size_t ws_bytes, float* loss_out, float* dloss_dimage,
                 void* stream);

/* ---- adaptive density control + optimizer surgery + Adam -----------
 * The six training tensors of reference gsplat/gsmodel.py:96-129 (row-major [N, w], w =
 * 3, 3, high_sh_width (45 in the reference), 1, 3, 4); the same struct describes their Adam
 * moments (torch.optim.Adam state "exp_avg" / "exp_avg_sq"). */
typedef struct EgsGaussianParams {
  float* pws;
  float* low_shs;
  float* high_shs;
  float* alphas_raw;
  float* scales_raw;
  float* rots_raw;
} EgsGaussianParams;

/* GSModel.update_density_info (gsmodel.py:214-230): g = |dloss_dus[i]|; first != 0: grad_accum = g for
 * every row and count = visible; else rows with visible != 0 get grad_accum += g, count += 1.
 * visible: one byte per Gaussian (the `depths > 0.2` mask GSFunction returns). */
int egs_density_accumulate(int n, const float* dloss_dus, const uint8_t* visible, int first, float* grad_accum,
                           int32_t* count, void* stream);

/* GSModel.update_gaussian_density (gsmodel.py:232-317) in two calls around one 16-byte read-back.
 * plan: cls[i] = 0 pruned (alpha_raw < alpha_thr_raw or max scale_raw > big_thr_raw), 1 survivor,
 *   2 survivor + clone, 3 survivor + split (grad_accum/count >= grad_thr with 0/0 = 0; clone iff
 *   exp(max scale_raw) <= scale_thr); totals (device int32[4]) = {survivors, clones, splits, pruned}.
 * apply: out* have n_keep + n_clone + n_split rows: [survivors | clones | split children], each group in
 *   input order (== prune_params 151-166 followed by update_params 132-148).  Appended rows hold
 *   logit(sigmoid(alpha_raw)), log(exp(scale_raw) [* 0.6 for a split child]), the normalised quaternion,
 *   copied SH, pws [+ R(q) (exp(scale_raw) * z) for a split child]; their moments are zero.  The parent of
 *   a split stays unchanged (as in the reference).  z = unit_noise[3 * i + c] (i = INPUT row) when
 *   unit_noise != NULL, else a counter-based standard normal that is a pure function of
 *   (seed, round, i, c) -- identical on every replica (easygaussiansplatting_amd/scene.py:normal,
 *   stream = round, element 3 i + c).  in_exp_avg == NULL (with the other three moment sets) = the
 *   optimizer has no state yet.  `ws` is the workspace egs_densify_plan filled. */
size_t egs_densify_ws_bytes(int n);
int egs_densify_plan(int n, const float* alphas_raw, const float* scales_raw, const float* grad_accum,
                     const int32_t* count, float alpha_thr_raw, float big_thr_raw, float grad_thr, float scale_thr,
                     uint8_t* cls, void* ws, size_t ws_bytes, int32_t* totals, void* stream);
int egs_densify_apply(int n, int n_keep, int n_clone, int n_split, int high_sh_width, const uint8_t* cls,
                      const void* ws, const EgsGaussianParams* in, const EgsGaussianParams* in_exp_avg,
                      const EgsGaussianParams* in_exp_avg_sq, const EgsGaussianParams* out,
                      const EgsGaussianParams* out_exp_avg, const EgsGaussianParams* out_exp_avg_sq,
                      const float* unit_noise, uint64_t seed, uint64_t round, void* stream);

/* GSModel.reset_alpha (gsmodel.py:319-330): alphas_raw = min(alphas_raw, raw_val); both moments
 * (nullable) zeroed. */
int egs_reset_alpha(int n, float raw_val, float* alphas_raw, float* exp_avg, float* exp_avg_sq, void* stream);

/* torch.optim.Adam.step as the reference configures it (train.py:32; no weight decay / amsgrad):
 * all parameter groups in one launch.  step = the 1-based step count of the group AFTER this update.
 * betas/eps are doubles because torch derives 1 - beta and the bias corrections in double. */
typedef struct EgsAdamGroup {
  float* param;
  const float* grad;
  float* exp_avg;
  float* exp_avg_sq;
  int64_t count; /* floats */
  float lr;
  int32_t step;
} EgsAdamGroup;
int egs_adam_step(int n_groups, const EgsAdamGroup* groups, double beta1, double beta2, double eps, void* stream);
/* The same update for the SH coefficients of a step whose SH gradient stayed factored (EGS_BWD_FACTORED_SH): equal to
 * egs_sh_grad_views(n, sh_dim, views, pws, rows, row_stride, scale, grad..) followed by egs_adam_step on that gradient,
 * without the gradient rows ever reaching HBM.  low: the [n][3] tensor of the raw layout (gsmodel.py:117: low_shs) --
 * or, with high == NULL, the whole [n][sh_dim] tensor; high: the [n][sh_dim - 3] tensor (high_shs).  `grad` of the
 * groups is ignored, `count` must be n times the width. */
int egs_adam_sh_factored(int n, int sh_dim, int views, const float* pws, const float* rows, int64_t row_stride,
                         float scale, const EgsAdamGroup* low, const EgsAdamGroup* high /*nullable*/, double beta1,
                         double beta2, double eps, void* stream);

/* ---- initial scales from a point cloud -----------------------------
 * out_sqdist[i] = min_{j != i} |points[i] - points[j]|^2 (FLT_MAX when n == 1): the exact squared
 * distance to the nearest other point, i.e. faiss.IndexFlatL2(3).search(pws, 2)[0][:, 1] of
 * reference gsplat/read_write_model.py:216-220 (duplicates give 0).  points: [n,3]. */
size_t egs_nn_sqdist_ws_bytes(int n);
int egs_nn_sqdist(int n, const float* points, void* ws, size_t ws_bytes, float* out_sqdist, void* stream);

/* ---- the viewer's per-frame preprocess (last item) -------------------
 * reference viewer/shaders/gau_prep.glsl (dispatched by viewer/custom_items/gaussian_item.py:264-272):
 * gs_data [n, 11 + sh_dim] = {pos 3, rot 4, scale 3, alpha, sh} -> gs_prep [n,12] = {u 3 (NDC), covinv 3,
 * color 3, area 2, alpha} and depth [n] (view-space z, the viewer's sort key).  Culled rows (|u.xy| > 1.3,
 * |u.z| > 1, det == 0) only get u = -100; the rest of the row is left untouched, as the shader does.
 * view_matrix / projection_matrix: HOST float[16], row-major, mathematical convention (pc = V pw). */
int egs_viewer_prep(int n, int sh_dim, const float* gs_data, const float* view_matrix,
                    const float* projection_matrix, float focal_x, float focal_y, float* gs_prep, float* depth,
                    void* stream);

/* ---- per-kernel timing with HIP events on the launch stream ---------------
 * bench.py's `roofline` leg: when enabled, every kernel launch of this library
 * is bracketed by hipEventRecord on the stream it is launched on.
 * egs_prof_report() synchronises the recorded events and writes one line per
 * kernel name: "<name> <launches> <total_ms>\n"; returns the number of bytes
 * needed (excluding the NUL).  Recording every launch costs ~0.2 ms per 35-launch step on MI355X
 * (hipEventRecord serialises dispatch), hence the filter: bench.py times only the dominant kernel
 * inside its timed region. */
int egs_prof_enable(int on);
void egs_prof_set_filter(const char* kernel_name); /* NULL or "" = all kernels; else only that kernel */
void egs_prof_reset(void);
int egs_prof_report(char* buf, size_t cap);

#ifdef __cplusplus
}
#endif
#endif /* EGS_HIP_H_ */
