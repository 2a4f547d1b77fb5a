// Error reporting, policy presets and ABI version of libegs_hip.so.
#include "egs_common.h"

#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace egs {
static thread_local char g_err[512] = "no error";

void set_error(int code, const char* what, const char* file, int line) {
  const char* base = strrchr(file, '/');
  snprintf(g_err, sizeof(g_err), "egs error %d: %s (%s:%d)", code, what ? what : "?", base ? base + 1 : file, line);
}

// ---- per-kernel timing ------------------------------------------------------
struct ProfRec {
  const char* name;
  hipEvent_t a, b;
};
static std::mutex g_prof_mu;
static bool g_prof_enabled = false;
static std::vector<ProfRec> g_prof_recs;
static std::vector<hipEvent_t> g_prof_pool;
static std::string g_prof_filter;  // empty = every kernel

static hipEvent_t prof_event() {
  if (!g_prof_pool.empty()) {
    hipEvent_t e = g_prof_pool.back();
    g_prof_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}

bool prof_on(const char* name) {
  if (!g_prof_enabled) return false;
  return g_prof_filter.empty() || g_prof_filter == name;
}

void prof_begin(const char* name, hipStream_t s) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  ProfRec r{name, prof_event(), prof_event()};
  (void)hipEventRecord(r.a, s);
  g_prof_recs.push_back(r);
}

void prof_end(hipStream_t s) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (!g_prof_recs.empty()) (void)hipEventRecord(g_prof_recs.back().b, s);
}
}  // namespace egs

extern "C" int egs_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(egs::g_prof_mu);
  const int prev = egs::g_prof_enabled;
  egs::g_prof_enabled = on != 0;
  return prev;
}

extern "C" void egs_prof_set_filter(const char* kernel_name) {
  std::lock_guard<std::mutex> lk(egs::g_prof_mu);
  egs::g_prof_filter = kernel_name ? kernel_name : "";
}

extern "C" void egs_prof_reset(void) {
  std::lock_guard<std::mutex> lk(egs::g_prof_mu);
  for (auto& r : egs::g_prof_recs) {
    egs::g_prof_pool.push_back(r.a);
    egs::g_prof_pool.push_back(r.b);
  }
  egs::g_prof_recs.clear();
}

extern "C" int egs_prof_report(char* buf, size_t cap) {
  std::lock_guard<std::mutex> lk(egs::g_prof_mu);
  std::map<std::string, std::pair<int, double>> agg;
  for (auto& r : egs::g_prof_recs) {
    float ms = 0.f;
    if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
      auto& e = agg[r.name];
      e.first += 1;
      e.second += ms;
    }
  }
  std::string out;
  char line[256];
  for (auto& kv : agg) {
    snprintf(line, sizeof(line), "%s %d %.6f\n", kv.first.c_str(), kv.second.first, kv.second.second);
    out += line;
  }
  if (buf && cap > 0) {
    const size_t nb = out.size() < cap - 1 ? out.size() : cap - 1;
    memcpy(buf, out.data(), nb);
    buf[nb] = 0;
  }
  return (int)out.size();
}

// ---- HBM bandwidth probe: the device-to-device copy bench.py calibrates its roofline peak with --------
namespace egs {
typedef float f4v __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_hbm_copy(const f4v* __restrict__ src, f4v* __restrict__ dst, size_t n4) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride)
    __builtin_nontemporal_store(__builtin_nontemporal_load(&src[i]), &dst[i]);
}
}  // namespace egs

extern "C" int egs_hbm_copy_probe(void* dst, const void* src, size_t bytes, void* stream) {
  EGS_CHECK_ARG(dst && src && bytes >= 16 && (((uintptr_t)dst | (uintptr_t)src) & 15) == 0);
  const size_t n4 = bytes / 16;
  // 4 workgroups per CU, one float4 per lane and iteration, streaming (nt) loads and stores: the best of the
  // shapes tried on this pool (tools/ubench_copy.hip: 4.7-5.8 TB/s read + write; hipMemcpyAsync D2D 5.0;
  // MI355X_MICROARCH.md quotes 6.29 TB/s for its float4 copy)
  hipLaunchKernelGGL(egs::k_hbm_copy, dim3(256 * 4), dim3(256), 0, (hipStream_t)stream, (const egs::f4v*)src,
                     (egs::f4v*)dst, n4);
  EGS_LAUNCH_OK();
  return 0;
}

// ---- bitwise comparison of two device buffers (the public splat / splatB pair's content check, gsplatcu.py) -------------
namespace egs {
// flag |= 1 where a 32-bit word differs.  VEC: both buffers 16-B aligned (uint4 per lane and iteration, streaming loads:
// each byte is read once); otherwise word by word.  One non-returning store per wave that saw a difference.
typedef uint32_t uint4v __attribute__((ext_vector_type(4)));
template <bool VEC>
__global__ __launch_bounds__(256) void k_words_differ(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b,
                                                      size_t n, int32_t* __restrict__ flag) {
  const size_t stride = (size_t)gridDim.x * 256;
  uint32_t d = 0;
  if (VEC) {
    const uint4v* a4 = (const uint4v*)a;
    const uint4v* b4 = (const uint4v*)b;
    const size_t n4 = n / 4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
      const uint4v x = __builtin_nontemporal_load(&a4[i]), y = __builtin_nontemporal_load(&b4[i]);
      d |= (x.x ^ y.x) | (x.y ^ y.y) | (x.z ^ y.z) | (x.w ^ y.w);
    }
    const size_t t = 4 * n4 + (size_t)blockIdx.x * 256 + threadIdx.x;     // (the <= 3 words behind the last uint4)
    if (t < n) d |= a[t] ^ b[t];
  } else {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) d |= a[i] ^ b[i];
  }
  if (__any(d != 0) && (threadIdx.x & 63) == 0) *flag = 1;
}
}  // namespace egs

extern "C" int egs_words_differ(const void* a, const void* b, int64_t n_words, int32_t* flag, void* stream) {
  EGS_CHECK_ARG(flag && n_words >= 0 && (n_words == 0 || (a && b)) && ((((uintptr_t)a | (uintptr_t)b) & 3) == 0));
  if (n_words == 0) return 0;
  const bool vec = (((uintptr_t)a | (uintptr_t)b) & 15) == 0;
  const size_t per = vec ? 4 * 256 : 256;
  const int grid = (int)std::min<size_t>(256 * 8, ((size_t)n_words + per - 1) / per);
  if (vec)
    hipLaunchKernelGGL(egs::k_words_differ<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint32_t*)a,
                       (const uint32_t*)b, (size_t)n_words, flag);
  else
    hipLaunchKernelGGL(egs::k_words_differ<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint32_t*)a,
                       (const uint32_t*)b, (size_t)n_words, flag);
  EGS_LAUNCH_OK();
  return 0;
}

// ---- shader clock under load: what bench.py prices a VALU roofline against ---------------------------------------
namespace egs {
// Every wave runs a chain of dependent v_fma (all SIMDs issue, as under the draw kernels); lane 0 of workgroup 0 reads the
// shader-clock counter (s_memtime) and the constant 100-MHz counter (s_memrealtime) on both sides of it.
__global__ __launch_bounds__(256) void k_clock_probe(unsigned long long* __restrict__ out, int iters, float seed) {
  unsigned long long c0 = 0, r0 = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) { c0 = __builtin_readcyclecounter(); r0 = wall_clock64(); }
  float a = seed + (float)threadIdx.x, b = 1.0000001f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) a = fmaf(a, b, 1.0e-7f);
  }
  if (a == 12345.678f) out[4] = 1ull;      // (keeps the chain alive)
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    out[0] = c0; out[1] = __builtin_readcyclecounter(); out[2] = r0; out[3] = wall_clock64();
  }
}
}  // namespace egs

// out: 8 uint64 in device memory; after the stream has run it, shader MHz = (out[1] - out[0]) / (out[3] - out[2]) * 100
extern "C" int egs_clock_probe(void* out, int iters, void* stream) {
  EGS_CHECK_ARG(out && iters > 0 && ((uintptr_t)out & 7) == 0);
  hipLaunchKernelGGL(egs::k_clock_probe, dim3(256 * 8), dim3(256), 0, (hipStream_t)stream, (unsigned long long*)out,
                     iters, 0.5f);
  EGS_LAUNCH_OK();
  return 0;
}

// ---- mailbox: page-locked landing slots for the read-back of the enqueue-ahead path ------------------
namespace egs {
struct Mailbox {
  int slots;
  uint32_t* host;                 // slots x 4 words, page-locked, COHERENT (kernel stores are polled by the host)
  uint32_t* dev;                  // the same memory as the device addresses it
  std::vector<hipEvent_t> ev;     // recorded behind the copy into the slot (egs_mailbox_post)
  std::vector<char> armed;        // 1: the slot is filled by kernel stores and POLLED (egs_mailbox_arm)
  std::vector<hipStream_t> stream;  // ... by kernels of this stream (0 = the default stream)
};
constexpr uint32_t MAILBOX_EMPTY = 0xFFFFFFFFu;   // never a patch count (P < 2^31)
static inline void cpu_relax() {   // spin-wait hint of the host architecture
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#elif defined(__aarch64__) || defined(__arm__)
  __asm__ __volatile__("yield");
#else
  __asm__ __volatile__("" ::: "memory");
#endif
}
}  // namespace egs

extern "C" void* egs_mailbox_create(int slots) {
  if (slots <= 0 || slots > 4096) return nullptr;
  egs::Mailbox* m = new egs::Mailbox();
  m->slots = slots;
  m->host = nullptr;
  // coherent (fine-grained) and mapped, explicitly: the slots are written by kernel stores and polled by the host
  // while the stream is still running -- "default" host memory is only coherent as long as HIP_HOST_COHERENT says so
  if (hipHostMalloc((void**)&m->host, (size_t)slots * 16, hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) {
    (void)hipGetLastError();
    if (hipHostMalloc((void**)&m->host, (size_t)slots * 16, hipHostMallocDefault) != hipSuccess) {
      delete m;
      return nullptr;
    }
  }
  m->dev = nullptr;
  if (hipHostGetDevicePointer((void**)&m->dev, m->host, 0) != hipSuccess || !m->dev) {
    (void)hipGetLastError();
    m->dev = m->host;              // unified addressing: the host pointer is valid on the device
  }
  memset(m->host, 0xFF, (size_t)slots * 16);
  m->ev.resize(slots, nullptr);
  m->armed.resize(slots, 0);
  m->stream.resize(slots, nullptr);
  for (int i = 0; i < slots; ++i)
    if (hipEventCreateWithFlags(&m->ev[i], hipEventDisableTiming) != hipSuccess) {
      for (int j = 0; j < i; ++j) (void)hipEventDestroy(m->ev[j]);
      (void)hipHostFree(m->host);
      delete m;
      return nullptr;
    }
  return m;
}

extern "C" void egs_mailbox_destroy(void* mb) {
  egs::Mailbox* m = (egs::Mailbox*)mb;
  if (!m) return;
  for (auto e : m->ev) (void)hipEventDestroy(e);
  (void)hipHostFree(m->host);
  delete m;
}

extern "C" int egs_mailbox_post(void* mb, int slot, const uint32_t* total_patches, void* stream) {
  egs::Mailbox* m = (egs::Mailbox*)mb;
  EGS_CHECK_ARG(m && slot >= 0 && slot < m->slots && total_patches);
  hipStream_t s = (hipStream_t)stream;
  m->armed[slot] = 0;
  EGS_HIP(hipMemcpyAsync(m->host + 4 * (size_t)slot, total_patches, 8, hipMemcpyDeviceToHost, s));
  EGS_HIP(hipEventRecord(m->ev[slot], s));
  return 0;
}

extern "C" uint32_t* egs_mailbox_slot(void* mb, int slot) {
  egs::Mailbox* m = (egs::Mailbox*)mb;
  return (m && slot >= 0 && slot < m->slots) ? m->dev + 4 * (size_t)slot : nullptr;   // what the kernels store to
}

// Arm a slot for kernel stores: word 0 (the patch count, stored LAST in stream order by the binning kernels)
// is set to a value no patch count can take; egs_mailbox_fetch then polls it.  Nothing is enqueued -- an event
// record behind the binning stage was measured to open a 6 us bubble in front of the next kernel.
extern "C" int egs_mailbox_arm(void* mb, int slot, void* stream) {
  egs::Mailbox* m = (egs::Mailbox*)mb;
  EGS_CHECK_ARG(m && slot >= 0 && slot < m->slots);
  volatile uint32_t* h = m->host + 4 * (size_t)slot;
  h[1] = 0u;
  h[0] = egs::MAILBOX_EMPTY;
  __sync_synchronize();
  m->armed[slot] = 1;
  m->stream[slot] = (hipStream_t)stream;
  return 0;
}

extern "C" int egs_mailbox_fetch(void* mb, int slot, int blocking, uint32_t* out) {
  egs::Mailbox* m = (egs::Mailbox*)mb;
  if (!m || slot < 0 || slot >= m->slots || !out) {
    egs::set_error(EGS_ERR_BAD_ARG, "bad argument: mailbox / slot / out", __FILE__, __LINE__);
    return -EGS_ERR_BAD_ARG;
  }
  const volatile uint32_t* h = m->host + 4 * (size_t)slot;
  if (m->armed[slot]) {   // filled by kernel stores: poll word 0 (the max key, word 1, was stored by an EARLIER kernel)
    if (h[0] == egs::MAILBOX_EMPTY) {
      if (!blocking) return 0;
      const auto t0 = std::chrono::steady_clock::now();
      uint64_t spins = 0;
      while (h[0] == egs::MAILBOX_EMPTY) {
        egs::cpu_relax();
        if ((++spins & 0xFFF) == 0 &&
            std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) {   // (never seen) ask the runtime
          const hipError_t e = hipStreamSynchronize(m->stream[slot]);
          if (e != hipSuccess || h[0] == egs::MAILBOX_EMPTY) {
            egs::set_error((int)e, "mailbox slot was not written by the binning stage", __FILE__, __LINE__);
            return -(e != hipSuccess ? (int)e : EGS_ERR_BAD_ARG);
          }
        }
      }
    }
    __sync_synchronize();
    out[0] = h[0];
    out[1] = h[1];
    return 1;
  }
  hipError_t e = hipEventQuery(m->ev[slot]);
  if (e == hipErrorNotReady) {
    if (!blocking) return 0;
    e = hipEventSynchronize(m->ev[slot]);
  }
  if (e != hipSuccess) {
    egs::set_error((int)e, hipGetErrorString(e), __FILE__, __LINE__);
    return -(int)e;
  }
  out[0] = h[0];
  out[1] = h[1];
  return 1;
}

extern "C" int egs_mailbox_peek(void* mb, int slot, uint32_t* out4) {
  egs::Mailbox* m = (egs::Mailbox*)mb;
  EGS_CHECK_ARG(m && slot >= 0 && slot < m->slots && out4);
  const volatile uint32_t* h = m->host + 4 * (size_t)slot;
  for (int i = 0; i < 4; ++i) out4[i] = h[i];
  return 0;
}

extern "C" int egs_mailbox_clear(void* mb, int slot) {
  egs::Mailbox* m = (egs::Mailbox*)mb;
  EGS_CHECK_ARG(m && slot >= 0 && slot < m->slots);
  volatile uint32_t* h = m->host + 4 * (size_t)slot;
  for (int i = 0; i < 4; ++i) h[i] = egs::MAILBOX_EMPTY;
  __sync_synchronize();
  return 0;
}

extern "C" const char* egs_last_error_string(void) { return egs::g_err; }
extern "C" int egs_abi_version(void) { return EGS_ABI_VERSION; }

// The CUDA extension's semantics (reference gsplatcu/kernel.cu, gausplat.cu): drop-in default.
extern "C" void egs_policy_gsplatcu(EgsPolicy* p) {
  p->near_cull = 1;
  p->fov_mode = 0;
  p->det_eps = 0.f;
  p->nan_cull = 1;
  p->radius_mode = 0;
  p->footprint = 0;
  p->far_cull = 0;
  p->maha_floor = 1;
  p->alpha_clamp = 1;
  p->alpha_skip = 0.002f;
  p->tau_stop = 0.0001f;
  p->depth_key = 0;
  p->nan_maha = 0;
}

// forward_cpu.py semantics (reference gsplat/gausplat.py).
extern "C" void egs_policy_forward_cpu(EgsPolicy* p) {
  p->near_cull = 0;
  p->fov_mode = 1;
  p->det_eps = 0.000001f;
  p->nan_cull = 0;
  p->radius_mode = 1;
  p->footprint = 1;
  p->far_cull = 1;
  p->maha_floor = 0;
  p->alpha_clamp = 1;
  p->alpha_skip = 0.f;
  p->tau_stop = 0.f;
  p->depth_key = 1;
  p->nan_maha = 1;     // (numpy's exp(-0.5 * NaN) is NaN: gausplat.py has no defined result there; NaN pixels are skipped)
}
