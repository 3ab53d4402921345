// dfx_persist.hip -- the persistent stage-loop kernels of libdfx (dfx_persist.h), a translation unit of their own: see
// dfx_persist_api.h for why.  No host logic here beyond handing out kernel addresses.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>

#include "dfx_kernels.h"
#include "dfx_persist.h"

namespace dfx_persist {

template <int MODEL, int CONTACT>
static const void* fwd_kernel_t(int npb) {
  if (npb == 3) return (const void*)k_fwd_persist<MODEL, CONTACT, 3>;
  return (const void*)k_fwd_persist<MODEL, CONTACT, 4>;
}
const void* fwd_kernel(int model, int contact, int npb) {
  if (contact != 0 && contact != 1) return nullptr;
  if (model == kNonlinear) return contact ? fwd_kernel_t<kNonlinear, 1>(npb) : fwd_kernel_t<kNonlinear, 0>(npb);
  if (model == kLinearized) return contact ? fwd_kernel_t<kLinearized, 1>(npb) : fwd_kernel_t<kLinearized, 0>(npb);
  return nullptr;
}
template <int MODEL, int CONTACT>
static const void* adj_kernel_t(int npb) {
  if (npb == 3) return (const void*)k_adj_persist<MODEL, CONTACT, 3>;
  return (const void*)k_adj_persist<MODEL, CONTACT, 4>;
}
const void* adj_kernel(int model, int contact, int npb) {
  if (contact != 0 && contact != 1) return nullptr;
  if (model == kNonlinear) return contact ? adj_kernel_t<kNonlinear, 1>(npb) : adj_kernel_t<kNonlinear, 0>(npb);
  if (model == kLinearized) return contact ? adj_kernel_t<kLinearized, 1>(npb) : adj_kernel_t<kLinearized, 0>(npb);
  return nullptr;
}
void launch_ring_poison(hipStream_t st, double* ring, int batch, int n_blocks, int m0, int nm, int width) {
  const int per_member = n_blocks * width;
  hipLaunchKernelGGL(k_ring_poison, dim3((per_member + kThreads - 1) / kThreads, nm), dim3(kThreads), 0, st, ring, batch, n_blocks, m0, width, kPAhead);
}

}  // namespace dfx_persist
