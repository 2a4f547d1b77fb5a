// Error reporting, policy presets and ABI version of libegs_hip.so.
#include "egs_common.h"

#include <stdio.h>
#include <string.h>

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
}
