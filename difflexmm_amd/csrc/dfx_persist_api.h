// dfx_persist_api.h -- what the engine's host code and the persistent kernels' translation unit (dfx_persist.hip) share.
// The persistent kernels are compiled on their own because they need one compiler switch the stage kernels must not get:
// -mllvm -disable-machine-licm.  Their stage loop is the first long loop around the ligament arithmetic, and the machine-level
// loop-invariant code motion hoists every fp64 literal of that arithmetic (the polynomial coefficients of atan2 / sincos, ~80
// registers of v_mov) out of it: 203 VGPRs and two waves per SIMD instead of 108 and four (profiles/r05_persistent_kernels.txt).
#pragma once
#include <hip/hip_runtime.h>

namespace dfx_persist {

constexpr int kPRing = 8;      // places in the hand-off ring
constexpr int kPAhead = 4;     // a place is re-poisoned this many stage ordinals before its record is due
constexpr int kPersistThreads = 256;
constexpr unsigned kPoisonWord = 0xFFFFFFFFu;
constexpr int kPersistStages = 6;   // fixed-grid tableaus of the library: 4 (RK4) and 6 (Dormand-Prince)
constexpr int kSpinLimit = 1 << 23; // polls (>= 0.5 us each) before a wave gives up: seconds
// what the hand-off protocol leans on (dfx_persist.h): a place is re-poisoned kPAhead ordinals ahead of its record -- at least two, so that
// the store is complete (the owner's next poll waits for it) before a neighbour can ask -- and the ring is longer than that look-ahead
static_assert(kPAhead >= 2 && kPRing >= kPAhead + 1, "hand-off ring: re-poison at least two ordinals ahead, ring longer than the look-ahead");

struct PersistCoef {            // the whole tableau in acceleration form, by value (scalar loads at compile-time offsets)
  double cv[kPersistStages][kPersistStages];
  double cq[kPersistStages][kPersistStages];
  double c[kPersistStages + 1];
};
struct PersistArgs {
  double* ring;                 // kPRing * batch * n_blocks * kPos
  int* give_up;                 // pinned host word: != 0 once a wave gave up (1 + the stage ordinal it waited for)
  int n_steps, nm, waves_per_member;
  int spin_limit;               // polls before a wave gives up (kSpinLimit; the test hook dfx_test_set_spin_limit makes it tiny)
  int xcd_wg;                   // > 0: workgroups per member, every member on ONE XCD (workgroup b sits on XCD b % 8: persist_wave); 0: waves packed densely
#ifdef DFX_PERSIST_TIMING
  unsigned* dbg;                // diagnostic build: 8 words per wave (six phase sums, total ticks, stages)
#endif
};


struct PersistAdjCoef {         // AdjCoef of every stage
  double col[kPersistStages][kPersistStages + 1];
  double cur[kPersistStages][kPersistStages + 1];
  double c[kPersistStages];
};

// kernels by (bond model, contact, lanes per block); nullptr: no such build
const void* fwd_kernel(int model, int contact, int npb);
const void* adj_kernel(int model, int contact, int npb);
// the adaptive controller inside the stage loop and the reverse sweep of the steps it keeps (dfx_persist_dense.hip)
const void* adaptive_fwd_kernel(int model, int contact);
const void* adj_dense_kernel(int model, int contact, int npb);
// places 0 .. kPAhead-1 of the ring, members [m0, m0 + nm), poisoned on `st`
void launch_ring_poison(hipStream_t st, double* ring, int batch, int n_blocks, int m0, int nm, int width);

}  // namespace dfx_persist
