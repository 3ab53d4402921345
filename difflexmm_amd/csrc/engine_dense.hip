// engine_dense.hip -- libdfx host side: the reverse sweep of an adaptive solve that kept its accepted steps (dfx_forward_adaptive_keep): the exact
// discrete adjoint of those steps, the outputs' cotangents entering through the dense output (dfx_dense.h says how)
// (one of six translation units; shared declarations in dfx_engine.h, the design in DESIGN.md section 3)
#include "dfx_engine.h"
#include "dfx_dense.h"

using namespace dfx_persist;

template <int MODEL, int CONTACT>
static void launch_adj_dense_t(const DevCtx& c, hipStream_t st, dim3 grid, const AdjCoef& ac, const DenseCtx& dn, int i, int j) {
  if (c.g_b) hipLaunchKernelGGL((k_adj_stage_dense<MODEL, CONTACT, 1>), grid, dim3(kThreads), 0, st, c, ac, dn, i, j);
  else hipLaunchKernelGGL((k_adj_stage_dense<MODEL, CONTACT, 0>), grid, dim3(kThreads), 0, st, c, ac, dn, i, j);
}
static void launch_adj_dense(dfx_handle* h, const DevCtx& c, hipStream_t st, dim3 grid, const DenseCtx& dn, int i, int j) {
  const AdjCoef ac = adj_coef(h->pl.tab, i);
  const bool con = h->pl.contact != 0;
  if (h->pl.model == kNonlinear) { if (con) launch_adj_dense_t<kNonlinear, 1>(c, st, grid, ac, dn, i, j); else launch_adj_dense_t<kNonlinear, 0>(c, st, grid, ac, dn, i, j); }
  else { if (con) launch_adj_dense_t<kLinearized, 1>(c, st, grid, ac, dn, i, j); else launch_adj_dense_t<kLinearized, 0>(c, st, grid, ac, dn, i, j); }
  h->launches++;
}

int run_adjoint_dense(dfx_handle* h, const dfx_grads* want, dfx_grads* grads, dfx_grads* views, dfx_stats* stats, bool kinetic, int n_target) {
  const Plan& pl = h->pl;
  const size_t B = pl.batch, nb = pl.n_blocks;
  const int Tn = (int)h->ts.size(), s = pl.tab.s;
  DevCtx c = make_ctx(h);
  c.t_steps = h->d_tsteps.p; c.ts_stride = h->a_stride;          // every member its own step boundaries, as the controller left them
  c.fn_tab = pl.n_fns > 0 ? h->d_fn_tab.p : nullptr;             // (the DENSE builds read the segment's time-function table)
  h->launches = 0;
  h->persist_adj = false; h->pair_adj = false; h->lig_used = h->lig_adj_used = false;
  // steps 0 .. N_max of the longest member (N_max: the zero-size step of that member) in segments of <= kMaxGraphSteps
  const long long n_total = h->a_nmax + 1;
  h->n_total = n_total;
  h->segs.clear();
  for (long long base = 0; base < n_total; base += kMaxGraphSteps) {
    Seg sg;
    memset(&sg, 0, sizeof(sg));
    sg.base_step = base; sg.j0 = (int)std::min<long long>(base, 1 << 30); sg.interval = 0;
    sg.n_steps = (int)std::min<long long>(kMaxGraphSteps, n_total - base);
    h->segs.push_back(sg);
  }
  HIP_OK(h->d_segs.ensure(h->segs.size()));
  HIP_OK(hipMemcpyAsync(h->d_segs.p, h->segs.data(), sizeof(Seg) * h->segs.size(), hipMemcpyHostToDevice, h->stream));
  HIP_OK(hipMemsetAsync(h->d_YB.p, 0, sizeof(double) * B * s * nb * 6, h->stream));
  HIP_OK(hipMemsetAsync(h->d_LAM.p, 0, sizeof(double) * B * nb * 6, h->stream));
  HIP_OK(h->d_dw.ensure(B * (size_t)Tn * 8));
  HIP_OK(hipEventRecord(h->ev2, h->stream));
  DenseCtx dn;
  dn.out_ptr = h->d_out_ptr.p; dn.dw = h->d_dw.p; dn.n_acc = h->d_nacc.p; dn.stride = h->a_stride; dn.n_out = Tn; dn.pad = 0;
  const int total_w = (int)(B * (size_t)Tn);
  hipLaunchKernelGGL(k_dense_weights, dim3((total_w + kThreads - 1) / kThreads), dim3(kThreads), 0, h->stream, (const double*)h->d_theta.p, h->d_dw.p, total_w,
                     make_dopri());
  hipLaunchKernelGGL(k_adj_begin_dense, slot_grid(h), dim3(kThreads), 0, h->stream, c, dn);
  h->launches += 2;
  const dim3 grid = slot_grid(h);
  // one launch per segment where the sweep fits the chip at once (k_adj_dense_loop), one launch per stage otherwise
  if (ensure_flags(h)) return 2;
  *persist_give_up_word(h) = 0;
  const bool persist = persist_plan_adj_dense(h, c);
  for (int si = (int)h->segs.size() - 1; si >= 0; --si) {
    const Seg& sg = h->segs[si];
    hipLaunchKernelGGL(k_set_seg, dim3(1), dim3(1), 0, h->stream, (const Seg*)h->d_segs.p, si, h->d_cur.p);
    h->launches++;
    launch_fn_table(h, c, h->stream, (int)B, sg.n_steps);
    if (persist) { launch_adj_dense_persist(h, c, h->stream, sg.n_steps, dn); continue; }
    for (int j = sg.n_steps - 1; j >= 0; --j) {
      const long long n = sg.base_step + j;
      for (int i = (n == h->a_nmax ? 0 : s - 1); i >= 0; --i) launch_adj_dense(h, c, h->stream, grid, dn, i, j);
    }
  }
  if (kinetic) {
    dim3 g((unsigned)((n_target * 3 + 63) / 64), (unsigned)B);
    hipLaunchKernelGGL(k_kinetic_mass_grad, g, dim3(64), 0, h->stream, c, (const double*)h->d_fields.p, (const int32_t*)h->d_target.p, n_target);
  }
  HIP_OK(hipEventRecord(h->ev3, h->stream));
  if (int rc = collect_grads(h, want, grads, views, true)) return rc;
  if (*persist_give_up_word(h)) {
    // a workgroup of a persistent launch was not resident: one launch per stage from now on, and the sweep once more -- same process, same records
    persist_fell_back(h);
    if (zero_grad_accumulators(h, nullptr, 0, -1)) return 2;
    return run_adjoint_dense(h, want, grads, views, stats, kinetic, n_target);
  }
  if (stats) {
    memset(stats, 0, sizeof(*stats));
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, h->ev2, h->ev3);
    stats->steps = h->a_nmax;
    stats->rhs_evals = h->a_nmax * s;
    stats->launches = h->launches;
    stats->kernel_ms = ms;
    stats->streams = 1;
    stats->stage_kernel_us = h->a_nmax ? 1e3 * ms / (double)(h->a_nmax * s * 2) : 0.0;
    stats->checkpoint_records = 1;
    stats->tile_kernels = h->persist_adj ? 3 : 0;
  }
  return 0;
}
