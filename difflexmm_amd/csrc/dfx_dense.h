// dfx_dense.h -- the reverse sweep of an adaptive solve that kept its accepted steps (dfx_forward_adaptive_keep, include/dfx.h): the small
// kernels around the DENSE builds of the reverse stage (adj_stage_body<..., DENSE = 1> in dfx_kernels.h).
//
// The reference differentiates odeint (difflexmm/dynamics.py:166) with jax's continuous adjoint; here the reverse sweep is the exact
// discrete adjoint of the accepted steps with their sizes frozen.  The outputs of that solve are not step states: they are interpolated
// inside the steps by jax's quartic dense output, which is linear in the step's slopes,
//     out(r) = y_n + h_n sum_{j=0..6} B_j(r) k_j,      k_6 = f(y_{n+1})  (the FSAL slope: the first slope of step n + 1),
// so the cotangent g of an output at relative position r of step n adds g to lambda_n and h_n B_j(r) g to Kbar_j; the share of k_6 joins
// Kbar_0 of step n + 1 -- or, after the last step, one extra evaluation at the final state, run as the launch (N_m, 0) of a step of size
// zero.  Oracle: oracle/ref_dynamics.solve_adaptive_replay_differentiable (autograd through the replay of the accepted steps).
#pragma once
#include "dfx_kernels.h"

namespace {

// theta -> the seven slope weights of every output of every member
__global__ __launch_bounds__(kThreads) void k_dense_weights(const double* theta, double* dw, int total, Dopri D) {
  const int idx = blockIdx.x * kThreads + threadIdx.x;
  if (idx >= total) return;
  double B[7];
  dopri_dense_weights(theta[idx], D.a[6], D.cm, B);
  for (int j = 0; j < 7; ++j) dw[(size_t)idx * 8 + j] = B[j];
  dw[(size_t)idx * 8 + 7] = 0.0;
}

// Start of the sweep, per member: lambda and Ybar are zero (the prelude launch), and the (w, Kbar_q) input of the launch (N_m, 0) -- the
// evaluation at the final state -- is h_{N-1} B_6 g of the outputs inside the last step.
__global__ __launch_bounds__(kThreads) void k_adj_begin_dense(DevCtx c, DenseCtx dn) {
  const int m = blockIdx.y + c.m0;
  const int tid = blockIdx.x * kThreads + threadIdx.x;
  if (tid >= c.n_slots) return;
  const int b = tid >> 2, d = tid & 3;
  if (d == 3) return;
  const size_t nd = (size_t)c.n_blocks * 3, nd6 = (size_t)c.n_blocks * 6;
  const long long N = dn.n_acc[m];
  const int sidx = c.block_special[b];
  const bool con = sidx >= 0 && ((c.special[sidx].con_mask >> d) & 1);
  double kq = 0.0, kv = 0.0;
  if (N > 0 && !con) {
    const double* ts = steps_of(c, m);
    const int* op = dn.out_ptr + (size_t)m * dn.stride;
    const double h_last = ts[N] - ts[N - 1];
    for (int kk = op[N - 1]; kk < op[N]; ++kk) {
      const double* G = c.G + ((size_t)kk * c.batch + m) * nd6;
      const double w6 = dn.dw[((size_t)m * dn.n_out + kk) * 8 + 6];
      kq += w6 * G[b * 6 + d]; kv += w6 * G[b * 6 + 3 + d];
    }
    kq *= h_last; kv *= h_last;
  }
  const int win = (int)((N * c.s) & 1);             // parity of the launch (N, 0)
  c.KQ[((size_t)m * 2 + win) * nd + b * 3 + d] = kq;
  c.W[((size_t)m * 2 + win) * nd + b * 3 + d] = kv * c.inv_m[(size_t)m * nd + b * 3 + d];
}

// the DENSE builds of the reverse stage: records build (gradients a design reaches) and the build with per-ligament gradients
template <int MODEL, int CONTACT, int BOND_GRADS>
__global__ __launch_bounds__(kThreads) void k_adj_stage_dense(DevCtx c, AdjCoef ac, DenseCtx dn, int i, int j) {
  static_assert(kernel_takes_devctx_first(&k_adj_stage_dense<MODEL, CONTACT, BOND_GRADS>), "DevCtx must stay the first kernel argument");
  StageCoef rc;
  adj_stage_body<MODEL, CONTACT, BOND_GRADS, BOND_GRADS, 4, 1, 0, 0, -1, 1>(c, ac, i, j, -1 - i, -1, 0, rc, 0, dn);
}

}  // namespace
