// Error reporting, policy presets and ABI version of libegs_hip.so.
#include "egs_common.h"

#include <stdio.h>
#include <string.h>

namespace egs {
static thread_local char g_err[512] = "no error";

void set_error(int code, const char* what, const char* file, int line) {
  const char* base = strrchr(file, '/');
  snprintf(g_err, sizeof(g_err), "egs error %d: %s (%s:%d)", code, what ? what : "?", base ? base + 1 : file, line);
}
}  // namespace egs

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
