import os, subprocess, shutil
C='/root/repo/easygaussiansplatting_amd/csrc/'
def rep(s,a,b):
    assert s.count(a)==1,(s.count(a),a); return s.replace(a,b)
def mkpatch(name, edits, desc):
    tmp='/tmp/var/'+name; shutil.rmtree(tmp, ignore_errors=True); os.makedirs(tmp+'/a'); os.makedirs(tmp+'/b')
    out=desc
    for f,fn in edits.items():
        s=open(C+f).read(); open(tmp+'/a/'+f,'w').write(s); open(tmp+'/b/'+f,'w').write(fn(s))
        r=subprocess.run(['diff','-u','a/'+f,'b/'+f],cwd=tmp,capture_output=True,text=True).stdout
        r=r.replace('--- a/'+f,'--- a/easygaussiansplatting_amd/csrc/'+f).replace('+++ b/'+f,'+++ b/easygaussiansplatting_amd/csrc/'+f)
        import re
        r=re.sub(r'(^--- \S+)\t.*$', r'\1', r, flags=re.M); r=re.sub(r'(^\+\+\+ \S+)\t.*$', r'\1', r, flags=re.M)
        out+=r
    open('/root/repo/tools/lab/variants/'+name+'.patch','w').write(out)

# 1. issue-port probes of k_draw
def draw_probes(s):
    s=rep(s,'''  int gnext = (lane < n) ? gsid[r0 + lane] : 0;
  for (int base = 0; base < n && live != 0; base += 64) {
    __syncthreads();  // single-wave workgroup''','''  int gnext = (lane < n) ? gsid[r0 + lane] : 0;
#ifdef EGS_DRAW_DUMMY_SALU
  uint32_t dummy_s = 0;
#endif
#ifdef EGS_DRAW_DUMMY_VALU
  float dummy_v = 1.f;
#endif
  for (int base = 0; base < n && live != 0; base += 64) {
    __syncthreads();  // single-wave workgroup''')
    s=rep(s,'''        const float4 Q = sA[j], P = sB[j];            // wave-uniform address: LDS broadcast
        float4 K;
        if constexpr (BOX) K = sC[j];
        else { const float2 gb = *reinterpret_cast<const float2*>(&sC[j]); K = make_float4(P.w, gb.x, gb.y, 0.f); }
''','''#ifdef EGS_DRAW_PROBE_NOK    // LDS probe: two broadcast reads per entry instead of three (timing only: the colours are constants)
        const float4 Q = sA[j], P = sB[j], K = make_float4(0.5f, 0.25f, 0.125f, 0.f);
#else
        const float4 Q = sA[j], P = sB[j];            // wave-uniform address: LDS broadcast
        float4 K;
        if constexpr (BOX) K = sC[j];
        else { const float2 gb = *reinterpret_cast<const float2*>(&sC[j]); K = make_float4(P.w, gb.x, gb.y, 0.f); }
#endif
#ifdef EGS_DRAW_DUMMY_SALU   // issue-limit probe (tools/lab/lab_issue_probe.sh): N extra scalar instructions per entry
#pragma unroll
        for (int q = 0; q < EGS_DRAW_DUMMY_SALU; ++q) asm volatile("s_add_u32 %0, %0, 1" : "+s"(dummy_s) : : "scc");
#endif
#ifdef EGS_DRAW_DUMMY_VALU   // ... or N extra full-rate vector instructions
#pragma unroll
        for (int q = 0; q < EGS_DRAW_DUMMY_VALU; ++q) asm volatile("v_add_f32 %0, %0, %0" : "+v"(dummy_v));
#endif
''')
    s=rep(s,'''  if (p.work_out) {   // what k_draw_bwd will walk: the largest contributor index of the tile and of its blocks''','''#ifdef EGS_DRAW_DUMMY_SALU
  if (dummy_s == 0xFFFFFFFFu) cr[0] += 1.f;   // (keeps the probe's chain alive)
#endif
#ifdef EGS_DRAW_DUMMY_VALU
  if (dummy_v == 12345.f) cr[0] += 1.f;
#endif
  if (p.work_out) {   // what k_draw_bwd will walk: the largest contributor index of the tile and of its blocks''')
    return s
mkpatch('draw_issue_probes', {'egs_draw.hip': draw_probes}, '''Issue-port probes of k_draw (round 4, DESIGN 3.3 / LAB 3.3): -DEGS_DRAW_DUMMY_SALU=N / -DEGS_DRAW_DUMMY_VALU=N add N scalar /
vector instructions per (tile, entry); -DEGS_DRAW_PROBE_NOK reads two LDS pieces per entry instead of three and blends
constant colours (timing only, the image is not the scene's).  Apply, build a variant library, time it with
tools/lab/lab_issue_probe.sh.  Not part of the product sources.

''')

# 2. reduction probes + hit bits of k_draw_bwd
def bwd_probes(s):
    s=rep(s,'''__device__ __forceinline__ float rows_of4(float e0, float e1, float e2, float e3) {
''','''#ifndef EGS_PROBE_REDUCE   // timing probes (the sums are not the wave totals: gradients are not the scene's): 1 = the cross-row stage as plain adds, 2 = the in-row stage too
#define EGS_PROBE_REDUCE 0
#endif
__device__ __forceinline__ float rows_of4(float e0, float e1, float e2, float e3) {
#if EGS_PROBE_REDUCE
  return (e0 + e1) + (e2 + e3);
#endif
''')
    s=rep(s,'''__device__ __forceinline__ float rows_to_lanes9_bank(const float (&q)[9], int c16) {
''','''__device__ __forceinline__ float rows_to_lanes9_bank(const float (&q)[9], int c16) {
#if EGS_PROBE_REDUCE >= 2
  return ((q[0] + q[1]) + (q[2] + q[3])) + ((q[4] + q[5]) + (q[6] + q[7])) + q[8];
#endif
''')
    s=rep(s,'''      if (nanfix && !BOX && !p.masked) mymask = 0xF;
      sA[lane] = A;''','''      if (nanfix && !BOX && !p.masked) mymask = 0xF;
#if EGS_PROBE_HIT_BITS   // (measurement builds only: even this wave-uniform test cost the production kernel two spilled registers)
      if (p.hit_bits) {
        const uint32_t gi = (uint32_t)(r0 + idx);
        if (!((p.hit_bits[gi >> 5] >> (gi & 31u)) & 1u)) mymask = 0;
      }
#endif
      sA[lane] = A;''')
    s=rep(s,'''  p.masked = 0;
''','''  p.masked = 0;
  p.hit_bits = g_probe_hit_bits;
''')
    s=rep(s,'''DrawParams make_draw_params(int W, int H, const EgsPolicy* pol, bool backward) {''','''#ifndef EGS_PROBE_HIT_BITS
#define EGS_PROBE_HIT_BITS 0
#endif
// measurement probe (tools/lab/bwd_hit_stats.py --time): per-list-entry hit bits for the NEXT backward draws of this process
static const uint32_t* g_probe_hit_bits = nullptr;
extern "C" int egs_probe_set_hit_bits(const void* bits) {
  g_probe_hit_bits = (const uint32_t*)bits;
  return EGS_PROBE_HIT_BITS ? 0 : 1;
}
DrawParams make_draw_params(int W, int H, const EgsPolicy* pol, bool backward) {''')
    return s
def hdr_hit(s):
    return rep(s,'''  int masked;
};''','''  int masked;
  // k_draw_bwd only, measurement probe (egs_probe_set_hit_bits): one bit per list entry, 0 = the entry blended into no pixel
  // of its tile -- what a forward pass COULD leave behind; the backward pass then drops such entries before staging them
  const uint32_t* hit_bits;
};''')
mkpatch('draw_bwd_probes', {'egs_draw.hip': bwd_probes, 'egs_raster.h': hdr_hit}, '''Probes of k_draw_bwd (rounds 3-4, LAB 3.4): -DEGS_PROBE_REDUCE=1|2 replaces the transposing wave reduction by plain adds
(timing only); -DEGS_PROBE_HIT_BITS=1 + egs_probe_set_hit_bits() lets the kernel drop entries a forward pass could have
marked as hitting nothing (prices the "hit bit" proposal; tools/lab/bwd_hit_stats.py binds the symbol itself).
Not part of the product sources.

''')

