// stage_builds_adj.hip -- the per-stage builds of the reverse stage kernel (k_adj_stage<..., ISTAGE>: the launches that fill the chip, 2/3 of
// a C3 step), a translation unit of their own so that they can be compiled for what they are: kernels bound by the ISSUE of their ~300 fp64
// instructions per lane at four waves per SIMD.  Makefile: -mllvm -amdgpu-sched-strategy=max-ilp and DFX_ADJ_OCC = waves_per_eu(4) for this
// file only -- together 3 % on the launch (profiles/r05_reverse_stage_schedule.txt); the same strategy costs the forward kernel 6 %, and alone
// (without the occupancy pin) it costs this one 5 %.  It also takes these kernels out of a lottery: in the big translation unit their schedule
// depended on which OTHER kernels were compiled next to them (round 5's library split lost 3 % on them without touching a line of theirs).
// No host logic beyond the launch switch.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>

#include "dfx_kernels.h"

namespace dfx_hot {

template <int MODEL, int CONTACT, int NPB>
static bool adj_t(hipStream_t st, dim3 grid, const DevCtx& c, const AdjCoef& acf, int i, int j, int in_buf, int wbuf, int local_only,
                  const StageCoef& rc, int rb) {
#define DFX_ADJ_I(I) case I: hipLaunchKernelGGL((k_adj_stage<MODEL, CONTACT, 0, 0, NPB, 1, 0, 1, I>), grid, dim3(kThreads), 0, st, c, acf, i, j, in_buf, \
    wbuf, local_only, rc, rb); return true;
  switch (i) { DFX_ADJ_I(0) DFX_ADJ_I(1) DFX_ADJ_I(2) DFX_ADJ_I(3) DFX_ADJ_I(4) DFX_ADJ_I(5) default: break; }
#undef DFX_ADJ_I
  return false;
}
template <int MODEL, int CONTACT>
static bool adj_n(int npb, hipStream_t st, dim3 grid, const DevCtx& c, const AdjCoef& acf, int i, int j, int in_buf, int wbuf, int local_only,
                  const StageCoef& rc, int rb) {
  return npb == 3 ? adj_t<MODEL, CONTACT, 3>(st, grid, c, acf, i, j, in_buf, wbuf, local_only, rc, rb)
                  : adj_t<MODEL, CONTACT, 4>(st, grid, c, acf, i, j, in_buf, wbuf, local_only, rc, rb);
}
// false: no per-stage build for this (model, contact, lanes per block, stage) -- the caller launches the generic build
bool launch_adj_stage_build(int model, int contact, int npb, hipStream_t st, dim3 grid, const DevCtx& c, const AdjCoef& acf, int i, int j, int in_buf, int wbuf,
                            int local_only, const StageCoef& rc, int rb) {
  if (contact != 0 && contact != 1) return false;
  if (model == kNonlinear) return contact ? adj_n<kNonlinear, 1>(npb, st, grid, c, acf, i, j, in_buf, wbuf, local_only, rc, rb)
                                          : adj_n<kNonlinear, 0>(npb, st, grid, c, acf, i, j, in_buf, wbuf, local_only, rc, rb);
  if (model == kLinearized) return contact ? adj_n<kLinearized, 1>(npb, st, grid, c, acf, i, j, in_buf, wbuf, local_only, rc, rb)
                                           : adj_n<kLinearized, 0>(npb, st, grid, c, acf, i, j, in_buf, wbuf, local_only, rc, rb);
  return false;
}

}  // namespace dfx_hot
