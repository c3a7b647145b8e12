// Diagnostic scaffolding of the blend kernels, kept OUT of the product kernels' text: with
// -DMGS_STAMP (profiles/stamp_forward.py / stamp_backward.py build their own library that way)
// every workgroup records start / end times, phase times and visit counts; in the product build
// (no -DMGS_STAMP) every macro below is empty and nothing of this file reaches the code object.
// raster_forward.hip defines MGS_DIAG_FORWARD, raster_backward.hip MGS_DIAG_BACKWARD before including it.
#pragma once
#include <hip/hip_runtime.h>

namespace mgs {

// ---- forward (k_blend_fwd) ----------------------------------------------------------------------
#if defined(MGS_STAMP) && defined(MGS_DIAG_FORWARD)   // diagnostic build only (profiles/stamp_forward.py): per-workgroup start/end stamps
__device__ long long g_stamps[4 * 65536];
__device__ long long g_phase[4 * 65536];
extern "C" int mgs_debug_read_phases(long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_phase), sizeof(long long) * n);
}
extern "C" int mgs_debug_read_stamps(long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamps), sizeof(long long) * n);
}
struct StampScope {
  long long t0, c0;
  int nseg = 0, nvisit = 0;
  long long ph[4] = {0, 0, 0, 0}, tl = 0;
  __device__ void mark(int k) { const long long t = __builtin_amdgcn_s_memtime(); ph[k] += t - tl; tl = t; }
  __device__ StampScope() : t0(__builtin_amdgcn_s_memrealtime()), c0(__builtin_amdgcn_s_memtime()) { tl = c0; }
  __device__ ~StampScope() {
    if (threadIdx.x == 0 && blockIdx.x < 65536) {
      g_stamps[4 * blockIdx.x + 0] = t0;
      g_stamps[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
      g_stamps[4 * blockIdx.x + 2] = ((__builtin_amdgcn_s_memtime() - c0) & 0xFFFFFFll) | ((long long)nseg << 24) | ((long long)nvisit << 40);
      g_stamps[4 * blockIdx.x + 3] = ((long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) << 32) |
                                     __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
      for (int k = 0; k < 4; k++) g_phase[4 * blockIdx.x + k] = ph[k];
    }
  }
};
#define MGS_STAMP_SCOPE StampScope stamp_scope_
#define MGS_STAMP_SEG(m) (stamp_scope_.nseg++, stamp_scope_.nvisit += __popcll(m))
#define MGS_MARK(k) stamp_scope_.mark(k)
#else
#define MGS_MARK(k)
#define MGS_STAMP_SCOPE
#define MGS_STAMP_SEG(m)
#endif

// ---- backward (k_blend_bwd) ---------------------------------------------------------------------
#if defined(MGS_STAMP) && defined(MGS_DIAG_BACKWARD)   // diagnostic build only (profiles/stamp_backward.py): per-item start/end stamps and phase times
__device__ long long g_bstamps[4 * 65536];
__device__ long long g_bphase[4 * 65536];
extern "C" int mgs_debug_read_bwd_stamps(long long* stamps, long long* phases, int n) {
  const int rc = (int)hipMemcpyFromSymbol(stamps, HIP_SYMBOL(g_bstamps), sizeof(long long) * n);
  return rc ? rc : (int)hipMemcpyFromSymbol(phases, HIP_SYMBOL(g_bphase), sizeof(long long) * n);
}
__device__ int* g_item_order = nullptr;     // diagnostic: dispatch order of the items (experiment)
extern "C" int mgs_debug_set_item_order(int* dev_ptr) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_item_order), &dev_ptr, sizeof(int*));
}
struct BwdStamp {
  long long t0, tl, ph[4] = {0, 0, 0, 0};
  int nvisit = 0, nany = 0, item_id = -1, item_base = 0, nmiss = 0, nlanes = 0;
  __device__ BwdStamp() : t0(__builtin_amdgcn_s_memrealtime()), tl(__builtin_amdgcn_s_memtime()) {}
  __device__ void mark(int k) { const long long t = __builtin_amdgcn_s_memtime(); ph[k] += t - tl; tl = t; }
  __device__ ~BwdStamp() {
    if (threadIdx.x == 0 && blockIdx.x < 65536) {
      g_bstamps[4 * blockIdx.x + 0] = t0;
      g_bstamps[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
      g_bstamps[4 * blockIdx.x + 2] = ((long long)nany << 32) | nvisit;
#ifdef MGS_STAMP_PH0   // phase 0 (item start -> per-pixel state in registers) instead of the lane statistics
      g_bphase[4 * blockIdx.x + 0] = ph[0];
#else
      g_bphase[4 * blockIdx.x + 0] = ((long long)nmiss << 32) | (unsigned int)nlanes;
#endif
      g_bstamps[4 * blockIdx.x + 3] = ((long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) << 32) |
                                      __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
      for (int k = 1; k < 3; k++) g_bphase[4 * blockIdx.x + k] = ph[k];
#ifdef MGS_STAMP_FINE
      g_bphase[4 * blockIdx.x + 3] = ph[3];
#else
      g_bphase[4 * blockIdx.x + 3] = ((long long)item_base << 32) | (unsigned int)item_id;
#endif
    }
  }
};
#define MGS_BSTAMP BwdStamp bstamp_
#ifdef MGS_STAMP_FINE   // the item's prologue in four steps (item record | quadrant ends | per-pixel loads | rest of the state phase)
#define MGS_BMARK(k)
#define MGS_BFINE(k, wait) do { asm volatile(wait ::: "memory"); bstamp_.mark(k); } while (0)
#else
#define MGS_BMARK(k) bstamp_.mark(k)
#define MGS_BFINE(k, wait)
#endif
#define MGS_BCOUNT(v, a) (bstamp_.nvisit += (v), bstamp_.nany += (a))
#define MGS_BITEM(item, base) (bstamp_.item_id = (item), bstamp_.item_base = (base))
#define MGS_BLANES(k) (bstamp_.nmiss += (__ballot(k) == 0ull), bstamp_.nlanes += __popcll(__ballot(k)))
#define MGS_BORDER(first, sketch) do { if (g_item_order && !(sketch)) first = g_item_order[first]; } while (0)
#else
#define MGS_BSTAMP
#define MGS_BMARK(k)
#define MGS_BFINE(k, wait)
#define MGS_BCOUNT(v, a)
#define MGS_BITEM(item, base)
#define MGS_BLANES(k)
#define MGS_BORDER(first, sketch)
#endif

}  // namespace mgs
