// dfx_persist_dense.hip -- the adaptive controller inside the persistent stage loop and the reverse sweep of the steps it keeps
// (dfx_persist_dense.h), a translation unit of their own, compiled like dfx_persist.hip (no machine-level loop-invariant code motion:
// dfx_persist_api.h says why).  No host logic here beyond handing out kernel addresses.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>

#include "dfx_kernels.h"
#include "dfx_persist_dense.h"

namespace dfx_persist {

const void* adaptive_fwd_kernel(int model, int contact) {
  if (contact != 0 && contact != 1) return nullptr;
  if (model == kNonlinear) return contact ? (const void*)k_adaptive_fwd_loop<kNonlinear, 1> : (const void*)k_adaptive_fwd_loop<kNonlinear, 0>;
  if (model == kLinearized) return contact ? (const void*)k_adaptive_fwd_loop<kLinearized, 1> : (const void*)k_adaptive_fwd_loop<kLinearized, 0>;
  return nullptr;
}
template <int MODEL, int CONTACT>
static const void* adj_dense_kernel_t(int npb) {
  if (npb == 3) return (const void*)k_adj_dense_loop<MODEL, CONTACT, 3>;
  return (const void*)k_adj_dense_loop<MODEL, CONTACT, 4>;
}
const void* adj_dense_kernel(int model, int contact, int npb) {
  if (contact != 0 && contact != 1) return nullptr;
  if (model == kNonlinear) return contact ? adj_dense_kernel_t<kNonlinear, 1>(npb) : adj_dense_kernel_t<kNonlinear, 0>(npb);
  if (model == kLinearized) return contact ? adj_dense_kernel_t<kLinearized, 1>(npb) : adj_dense_kernel_t<kLinearized, 0>(npb);
  return nullptr;
}

}  // namespace dfx_persist
