// engine_adaptive.hip -- libdfx host side: the reference's own integrator semantics (jax.experimental.ode.odeint called at dynamics.py:166):
// dfx_forward_adaptive and its step records
// (one of five translation units; shared declarations in dfx_engine.h, the design in DESIGN.md section 3)
#include "dfx_engine.h"

using namespace dfx_persist;

// ---- the accepted steps kept for the reverse sweep (dfx_forward_adaptive_keep) -------------------------------------------------
// Room for `cap` steps per member: stage records in the trajectory checkpoint (records level: 6 per step + the final state), step
// boundaries and output pointers with two spare entries (the zero-size step after the last one).  Growing keeps what is there.
template <class T>
static hipError_t regrow(DevBuf<T>& buf, size_t count, hipStream_t st) {
  if (count <= buf.n && buf.p) return hipSuccess;
  T* q = nullptr;
  hipError_t e = hipMalloc((void**)&q, count * sizeof(T));
  if (e != hipSuccess) return e;
  if (buf.p) {
    e = hipMemcpyAsync(q, buf.p, buf.n * sizeof(T), hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(buf.p);
  }
  buf.p = q; buf.n = count;
  return e;
}
static const char* const kNoRoom = "forward_adaptive_keep: the stage records of the accepted steps do not fit the device (the two-pass form -- "
                                   "dfx_forward_adaptive, then dfx_forward_grid on its step boundaries -- has the segments level to fall back on)";
static int adaptive_room(dfx_handle* h, long long cap, bool keep_contents) {
  const Plan& pl = h->pl;
  const size_t B = pl.batch, rec = (size_t)pl.n_blocks * kStep;
  const size_t want = B * ((size_t)cap * pl.tab.s + 1) * rec;
  if ((double)B * ((double)cap * pl.tab.s + 1.0) >= 4294967296.0) { h->err = kNoRoom; return 5; }
  const double* t0 = h->ck->traj.p;
  hipError_t e = keep_contents ? regrow(h->ck->traj, want, h->stream) : h->ck->traj.ensure(want);
  if (h->ck->traj.p != t0) h->ck->writer = nullptr;
  if (e != hipSuccess) { (void)hipGetLastError(); h->err = kNoRoom; return 5; }
  const long long stride = cap + 2;
  if (stride > h->a_stride || !h->d_tsteps.p || !h->d_out_ptr.p || h->d_tsteps.n < B * (size_t)stride || h->d_out_ptr.n < B * (size_t)stride) {
    DevBuf<double> nt; DevBuf<int> no;
    if (nt.ensure(B * (size_t)stride) != hipSuccess || no.ensure(B * (size_t)stride) != hipSuccess) {
      (void)hipGetLastError(); nt.release(); no.release(); h->err = kNoRoom; return 5;
    }
    if (keep_contents && h->a_stride > 0) {
      HIP_OK(hipMemcpy2DAsync(nt.p, sizeof(double) * stride, h->d_tsteps.p, sizeof(double) * h->a_stride, sizeof(double) * h->a_stride, B,
                              hipMemcpyDeviceToDevice, h->stream));
      HIP_OK(hipMemcpy2DAsync(no.p, sizeof(int) * stride, h->d_out_ptr.p, sizeof(int) * h->a_stride, sizeof(int) * h->a_stride, B,
                              hipMemcpyDeviceToDevice, h->stream));
      HIP_OK(hipStreamSynchronize(h->stream));
    }
    h->d_tsteps.release(); h->d_out_ptr.release();
    h->d_tsteps = nt; h->d_out_ptr = no;
    h->a_stride = stride;
  }
  h->a_cap = cap;
  return 0;
}

// after the last attempt: what the solve leaves in the handle, the fields, the statistics
static int finish_adaptive(dfx_handle* h, const std::vector<Clock>& clk, bool keep, double* fields, dfx_stats* stats, bool persist) {
  const Plan& pl = h->pl;
  const size_t B = pl.batch, nb = pl.n_blocks;
  const int Tn = (int)h->ts.size();
  // a member that was flagged instead of failing the call: its history is NaN from row 1 on (0xFF bytes are a NaN), and the reverse
  // sweep skips it (N_m = -1: every launch returns at once for it; its gradients stay zero and the caller sees its status)
  for (size_t m = 0; m < B; ++m)
    if (h->member_status[m] && Tn > 1) HIP_OK(hipMemsetAsync(h->d_fields.p + (m * Tn + 1) * nb * 6, 0xFF, sizeof(double) * (size_t)(Tn - 1) * nb * 6, h->stream));
  if (keep) {
    std::vector<int> nacc(B);
    h->a_nmax = 0;
    for (size_t m = 0; m < B; ++m) {
      nacc[m] = h->member_status[m] ? -1 : (int)clk[m].accepted;
      h->a_nmax = std::max<long long>(h->a_nmax, std::max(0, nacc[m]));
    }
    HIP_OK(hipMemcpyAsync(h->d_nacc.p, nacc.data(), sizeof(int) * B, hipMemcpyHostToDevice, h->stream));
    HIP_OK(hipStreamSynchronize(h->stream));
  }
  if (fields) HIP_OK(hipMemcpyAsync(fields, h->d_fields.p, sizeof(double) * B * Tn * nb * 6, hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  HIP_OK(hipGetLastError());
  h->have_fields = true;
  h->adaptive = false;
  h->adaptive_records = keep;
  h->have_adaptive_record = true;
  h->accepted_per_member.assign(B, 0);
  for (size_t m = 0; m < B; ++m) h->accepted_per_member[m] = clk[m].accepted;
  if (stats) {
    memset(stats, 0, sizeof(*stats));
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, h->ev0, h->ev1);
    long long acc = 0, att = 0;
    for (size_t m = 0; m < B; ++m) { acc = std::max(acc, clk[m].accepted); att = std::max(att, clk[m].attempts); }
    stats->steps = acc;
    stats->rhs_evals = 6 * att + 2;
    stats->launches = h->launches;
    stats->kernel_ms = ms;
    stats->streams = (!persist && h->groups.size() > 1 && !h->adaptive_exec) ? (int64_t)h->groups.size() : 1;
    stats->stage_kernel_us = att ? 1e3 * ms / (double)(att * (persist ? 7 : 8)) : 0.0;      // (persistent loop: 6 records + the error gather per attempt)
    stats->checkpoint_records = keep ? 1 : 0;
    stats->tile_kernels = persist ? 3 : 0;
  }
  return 0;
}

// ---- adaptive forward (reference odeint semantics) ------------------------------------------------
static int forward_adaptive_impl(dfx_handle* h, const double* state0, const double* timepoints, int32_t n_timepoints,
                                 double rtol, double atol, int64_t max_attempts, bool keep, double* fields, dfx_stats* stats);

int dfx_forward_adaptive(dfx_handle* h, const double* state0, const double* timepoints, int32_t n_timepoints,
                         double rtol, double atol, int64_t max_attempts, double* fields, dfx_stats* stats) {
  return forward_adaptive_impl(h, state0, timepoints, n_timepoints, rtol, atol, max_attempts, false, fields, stats);
}
int dfx_forward_adaptive_keep(dfx_handle* h, const double* state0, const double* timepoints, int32_t n_timepoints,
                              double rtol, double atol, int64_t max_attempts, int32_t keep_trajectory, double* fields, dfx_stats* stats) {
  return forward_adaptive_impl(h, state0, timepoints, n_timepoints, rtol, atol, max_attempts, keep_trajectory != 0, fields, stats);
}

static int forward_adaptive_impl(dfx_handle* h, const double* state0, const double* timepoints, int32_t n_timepoints,
                                 double rtol, double atol, int64_t max_attempts, bool keep, double* fields, dfx_stats* stats) {
  HIP_OK(hipSetDevice(h->device));
  h->persist_fwd = false;
  if (!h->have_params) { h->err = "forward_adaptive: set_params first"; return 1; }
  if (state0) for (size_t i = 0; i < (size_t)h->pl.batch * h->pl.n_blocks * 6; ++i)
    if (!std::isfinite(state0[i])) { h->err = "forward_adaptive: state0 holds a non-finite value"; return 1; }
  if (n_timepoints < 1) { h->err = "forward_adaptive: need >= 1 timepoint"; return 1; }
  if (h->pl.tab.s != 6) { h->err = "forward_adaptive: the adaptive controller is defined for the dopri5 tableau"; return 1; }
  const Plan& pl = h->pl;
  const size_t B = pl.batch, nb = pl.n_blocks, nd = nb * 3;
  const int Tn = n_timepoints;
  const Dopri D = make_dopri();
  h->ts.assign(timepoints, timepoints + Tn);
  h->spis.clear(); h->n_total = 0;
  h->have_traj = false; h->have_fields = false;
  h->adaptive_records = false;
  h->adaptive = true; h->rtol = rtol; h->atol = atol;
  h->have_adaptive_record = false;
  if (ensure_work_buffers(h)) { h->adaptive = false; return 2; }
  const int kAttemptsPerGraph = 32;
  long long cap_fit = 0;
  if (keep) {
    // the records build of the reverse stage serves these (launch_adj_dense); everything else keeps the two-pass form
    if (pl.n_ovf || !(pl.model == kNonlinear || pl.model == kLinearized) || pl.contact == DFX_CONTACT_DISTANCE) {
      h->adaptive = false;
      h->err = "forward_adaptive_keep: nonlinear / linearised ligaments with or without angle contact, one ligament per node (others: dfx_forward_adaptive, "
               "then dfx_forward_grid on its step boundaries)";
      return 5;
    }
    // room: what is free next to what this handle already holds, half of it at most (the reverse sweep's buffers come later), 5 % of the
    // device left alone; start with a few thousand steps and grow while the solve runs.  DFX_ADAPTIVE_CAP=n: start with n steps (tests)
    size_t free_b = 0, total_b = 0;
    HIP_OK(hipMemGetInfo(&free_b, &total_b));
    const double per_step = (double)pl.batch * pl.tab.s * pl.n_blocks * kStep * sizeof(double);
    const double usable = (double)free_b + (double)h->ck->traj.n * sizeof(double) - (double)total_b / 20.0;
    cap_fit = (long long)std::min(1048000.0, std::max(0.0, 0.5 * usable / per_step));
    long long cap = std::min<long long>(cap_fit, 4096);
    if (const char* e = getenv("DFX_ADAPTIVE_CAP")) cap = std::min<long long>(cap_fit, std::max<long long>(kAttemptsPerGraph + 8, atoll(e)));
    if (cap < kAttemptsPerGraph + 8) { h->adaptive = false; h->err = kNoRoom; return 5; }
    if (int rc = adaptive_room(h, cap, false)) { h->adaptive = false; return rc; }
    HIP_OK(h->d_theta.ensure(B * (size_t)n_timepoints));
    HIP_OK(h->d_nacc.ensure(B));
    h->have_traj = true; h->records = true; h->dense = false; h->segments = false; h->seg_chunk = 0;
    h->ck->writer = h;
    // rows of a solve without any step (one timepoint): the zero-size step at t_0 with the initial state as the only output
    std::vector<double> t_init(2 * B, timepoints[0]);
    std::vector<int> o_init(2 * B, 1);
    HIP_OK(hipMemcpy2DAsync(h->d_tsteps.p, sizeof(double) * h->a_stride, t_init.data(), sizeof(double) * 2, sizeof(double) * 2, B, hipMemcpyHostToDevice,
                            h->stream));
    HIP_OK(hipMemcpy2DAsync(h->d_out_ptr.p, sizeof(int) * h->a_stride, o_init.data(), sizeof(int) * 2, sizeof(int) * 2, B, hipMemcpyHostToDevice, h->stream));
    HIP_OK(hipStreamSynchronize(h->stream));      // (the two host vectors go out of scope)
  }
  const int n_wg = (pl.n_slots + kThreads - 1) / kThreads;
  const int n_partials = (pl.n_slots + 63) / 64;
  HIP_OK(h->d_fields.ensure(B * Tn * nb * 6));
  HIP_OK(h->d_clock.ensure(B));
  HIP_OK(h->d_err_partial.ensure(B * n_wg * kWavesPerWg));
  h->n_counts = std::max(0, Tn - 1);
  HIP_OK(h->d_step_counts.ensure(std::max<size_t>(1, B * h->n_counts)));
  HIP_OK(hipMemsetAsync(h->d_step_counts.p, 0, sizeof(int) * std::max<size_t>(1, B * h->n_counts), h->stream));
  HIP_OK(h->d_acc_times.ensure(B * (size_t)kAccCap));
  HIP_OK(h->d_ts.ensure(Tn));
  HIP_OK(h->d_tmp.ensure(std::max<size_t>(B * nb * 6, B)));
  HIP_OK(hipMemcpyAsync(h->d_ts.p, timepoints, sizeof(double) * Tn, hipMemcpyHostToDevice, h->stream));
  std::vector<double> rest;
  if (!state0) { rest.assign(B * nb * 6, 0.0); state0 = rest.data(); }      // NULL = every member starts at rest, as in dfx_forward
  HIP_OK(hipMemcpyAsync(h->d_state0.p, state0, sizeof(double) * B * nb * 6, hipMemcpyHostToDevice, h->stream));
  // constrained flags and free-DOF count
  std::vector<char> con(nd, 0);
  size_t n_free = 0;
  for (size_t b = 0; b < nb; ++b) {
    const int sidx = pl.block_special[b];
    for (int d = 0; d < 3; ++d) { con[b * 3 + d] = sidx >= 0 && ((pl.special[sidx].con_mask >> d) & 1); n_free += !con[b * 3 + d]; }
  }
  if (n_free == 0) { h->err = "forward_adaptive: no free DOF"; return 1; }
  // clocks for the two probing evaluations: h = 0, t = t0
  std::vector<Clock> clk(B);
  for (auto& c0 : clk) { memset(&c0, 0, sizeof(Clock)); c0.t = timepoints[0]; c0.out_idx = 1; }
  HIP_OK(hipMemcpyAsync(h->d_clock.p, clk.data(), sizeof(Clock) * B, hipMemcpyHostToDevice, h->stream));
  DevCtx c = make_ctx(h);
  h->launches = 0;
  dim3 g3((unsigned)((nb * 3 + kThreads - 1) / kThreads), (unsigned)B);
  hipLaunchKernelGGL(k_init, slot_grid(h), dim3(kThreads), 0, h->stream, c, (const double*)h->d_state0.p, timepoints[0], 0, 0LL, 0LL);
  if (keep)
    hipLaunchKernelGGL(k_checkpoint0, dim3((unsigned)((nb * kStep + kThreads - 1) / kThreads), (unsigned)B), dim3(kThreads), 0, h->stream, c, 0LL);
  hipLaunchKernelGGL(k_snapshot, g3, dim3(kThreads), 0, h->stream, c, h->d_fields.p, 0, h->d_seg_idx.p + 1, 0, 0LL, (int*)nullptr);
  launch_fwd(h, c, 0, 0, 0, -1, 0, 0);                      // A_0 = f(y0, t0)
  // only the evaluation just made comes back (row 0 / row 1 of every member's seven stage accelerations), not all of d_A
  std::vector<double> A0((size_t)B * nd), A1((size_t)B * nd);
  HIP_OK(hipMemcpy2DAsync(A0.data(), sizeof(double) * nd, h->d_A.p, sizeof(double) * 7 * nd, sizeof(double) * nd, B,
                          hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  // initial step size (Hairer II.4 as restated by jax, order 4), per member
  std::vector<double> y1(B * 2 * nd), tm(B), h0(B), d1v(B);
  for (size_t m = 0; m < B; ++m) {
    const double* q = state0 + m * 2 * nd; const double* v = q + nd; const double* a = A0.data() + m * nd;
    double d0 = 0, d1 = 0;
    for (size_t i = 0; i < nd; ++i) if (!con[i]) {
      const double sq = atol + fabs(q[i]) * rtol, sv = atol + fabs(v[i]) * rtol;
      d0 += (q[i] / sq) * (q[i] / sq) + (v[i] / sv) * (v[i] / sv);
      d1 += (v[i] / sq) * (v[i] / sq) + (a[i] / sv) * (a[i] / sv);
    }
    d0 = sqrt(d0); d1 = sqrt(d1);
    h0[m] = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
    d1v[m] = d1;
    for (size_t i = 0; i < nd; ++i) { y1[m * 2 * nd + i] = q[i] + h0[m] * (con[i] ? 0.0 : v[i]); y1[m * 2 * nd + nd + i] = v[i] + h0[m] * a[i]; }
    tm[m] = timepoints[0] + h0[m];
    clk[m].t = tm[m];
  }
  HIP_OK(hipMemcpyAsync(h->d_clock.p, clk.data(), sizeof(Clock) * B, hipMemcpyHostToDevice, h->stream));
  HIP_OK(hipMemcpyAsync(h->d_tmp.p, y1.data(), sizeof(double) * y1.size(), hipMemcpyHostToDevice, h->stream));
  DevBuf<double> d_tm;
  HIP_OK(d_tm.ensure(B));
  HIP_OK(hipMemcpyAsync(d_tm.p, tm.data(), sizeof(double) * B, hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(k_init_tm, slot_grid(h), dim3(kThreads), 0, h->stream, c, (const double*)h->d_tmp.p, (const double*)d_tm.p, 1);
  launch_fwd(h, c, 1, 0, 1, -1, 0, 0);                      // A_1 = f(y0 + h0 f0, t0 + h0)
  HIP_OK(hipMemcpy2DAsync(A1.data(), sizeof(double) * nd, h->d_A.p + nd, sizeof(double) * 7 * nd, sizeof(double) * nd, B,
                          hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  d_tm.release();
  for (size_t m = 0; m < B; ++m) {
    const double* q = state0 + m * 2 * nd; const double* v = q + nd;
    const double* a0 = A0.data() + m * nd; const double* a1 = A1.data() + m * nd; const double* v1 = y1.data() + m * 2 * nd + nd;
    double d2 = 0;
    for (size_t i = 0; i < nd; ++i) if (!con[i]) {
      const double sq = atol + fabs(q[i]) * rtol, sv = atol + fabs(v[i]) * rtol;
      const double x = (v1[i] - v[i]) / sq, y = (a1[i] - a0[i]) / sv;
      d2 += x * x + y * y;
    }
    d2 = sqrt(d2) / h0[m];
    const double h1 = (d1v[m] <= 1e-15 && d2 <= 1e-15) ? std::max(1e-6, h0[m] * 1e-3) : pow(0.01 / (d1v[m] + d2), 1.0 / 5.0);
    memset(&clk[m], 0, sizeof(Clock));
    clk[m].t = timepoints[0]; clk[m].t_last = timepoints[0]; clk[m].h = std::min(100.0 * h0[m], h1); clk[m].out_idx = 1;
    if (Tn == 1) clk[m].state = 1;
  }
  HIP_OK(hipMemcpyAsync(h->d_clock.p, clk.data(), sizeof(Clock) * B, hipMemcpyHostToDevice, h->stream));
  // coefficients
  DenseCoef dc;
  for (int l = 0; l < 7; ++l) { dc.cm[l] = D.cm[l]; dc.cma[l] = D.cma[l]; }
  dc.a10 = D.a[1][0];
  StageCoef sc_err;
  memset(&sc_err, 0, sizeof(sc_err));
  for (int l = 0; l < 7; ++l) { sc_err.cv[l] = D.e[l]; sc_err.cq[l] = D.ee[l]; }
  sc_err.c_i = 1.0; sc_err.c_next = 1.0;
  // one of the eight launches of an attempt (p = 0..4: evaluations at S_1..S_5, the last one leaves the candidate y1 in buffer 3;
  // 5: the FSAL evaluation with the error estimate; 6: controller; 7: dense output / commit / next stage-1 record) for the members of
  // one context (the whole batch, or one member group on its own stream)
  // keep: the attempt works in the records of step `accepted` of the trajectory checkpoint (record p + 1 -> p + 2; the candidate y1 is
  // record 6 = record 0 of the next step), and the controller records step boundaries and output positions (AdaptRec)
  AdaptRec ar;
  memset(&ar, 0, sizeof(ar));
  auto refresh_ar = [&]() { if (keep) { ar.t_steps = h->d_tsteps.p; ar.out_ptr = h->d_out_ptr.p; ar.theta = h->d_theta.p; ar.stride = h->a_stride; } };
  refresh_ar();
  auto launch_phase = [&](int p, const DevCtx& cc, hipStream_t st, dim3 grid, unsigned nm) {
    static const int inb[6] = {0, 1, 2, 1, 2, 1}, outb[6] = {0, 2, 1, 2, 1, 3};
    if (p < 5) { launch_fwd(h, cc, st, grid, p + 1, 0, keep ? -1 - (p + 1) : inb[p + 1], keep ? -2 - (p + 1) : outb[p + 1], keep ? -1 : 0, 0); return; }
    if (p == 5) {
      const int eb = keep ? -7 : 3, yb = keep ? -1 : 0;
#define DFX_ERR_CASE(M) case M: if (pl.contact == 2) hipLaunchKernelGGL((k_fwd_stage<M, 2>), grid, dim3(kThreads), 0, st, cc, sc_err, 6, 0, eb, -1, yb, 2); \
    else if (pl.contact) hipLaunchKernelGGL((k_fwd_stage<M, 1>), grid, dim3(kThreads), 0, st, cc, sc_err, 6, 0, eb, -1, yb, 2); \
    else hipLaunchKernelGGL((k_fwd_stage<M, 0>), grid, dim3(kThreads), 0, st, cc, sc_err, 6, 0, eb, -1, yb, 2); break;
      switch (pl.model) { DFX_ERR_CASE(kNonlinear) DFX_ERR_CASE(kLinearized) DFX_ERR_CASE(kSimpleSpring) DFX_ERR_CASE(kStretchTorsion) }
#undef DFX_ERR_CASE
    } else if (p == 6) hipLaunchKernelGGL(k_control, dim3(nm), dim3(kThreads), 0, st, cc, n_partials, 2.0 * (double)n_free, Tn, ar);
    else hipLaunchKernelGGL(k_prepare, grid, dim3(kThreads), 0, st, cc, dc, Tn, keep ? 1 : 0);
    h->launches++;
  };
  auto enqueue_attempt = [&]() { for (int p = 0; p < 8; ++p) launch_phase(p, c, h->stream, slot_grid(h), (unsigned)B); };
  HIP_OK(hipEventRecord(h->ev0, h->stream));
  // what the clocks say after a round of attempts: a member whose error estimate is not finite or whose step size underflowed stops
  // (its kernels return at once from then on); it fails the call -- or, dfx_set_failure_policy(h, 1), is flagged and the others go on
  h->member_status.assign(B, 0);
  auto judge = [&](bool& all_done) -> int {
    int rcj = 0;
    all_done = true;
    for (size_t m = 0; m < B; ++m) {
      if (clk[m].state == 2 || clk[m].state == 3) {
        h->member_status[m] = clk[m].state == 2 ? 1 : 2;
        if (!h->isolate_failures) {
          h->err = std::string("forward_adaptive: ") + (clk[m].state == 2 ? "non-finite error estimate" : "step size underflow") + " (member " + std::to_string(m) +
                   "; dfx_set_failure_policy(h, 1) flags such members instead of failing the call)";
          rcj = 3;
        }
      }
      if (clk[m].state == 0) all_done = false;
    }
    return rcj;
  };
  auto out_of_budget = [&]() -> int {
    int first = -1;
    for (size_t m = 0; m < B; ++m) if (clk[m].state == 0) { h->member_status[m] = 3; if (first < 0) first = (int)m; }
    if (h->isolate_failures) return 0;
    h->err = "forward_adaptive: step budget exceeded (member " + std::to_string(std::max(first, 0)) + " is not done after " + std::to_string((long long)max_attempts) +
             " attempts; dfx_set_failure_policy(h, 1) flags such members instead of failing the call)";
    return 4;
  };
  // ---- the controller inside the persistent stage loop (dfx_persist_dense.h) where the solve fits the chip at once: one launch carries
  // every member through up to kLoopAttempts attempts; the host only looks at the clocks between launches (and grows the room for kept steps)
  if (ensure_flags(h)) return 2;
  *persist_give_up_word(h) = 0;
  if (persist_adaptive_plan(h)) {
    int kLoopAttempts = 8192;
    if (const char* e = getenv("DFX_ADAPTIVE_LOOP_ATTEMPTS")) kLoopAttempts = std::max(1, atoi(e));
    int rcl = 0;
    while (true) {
      AdaptLoopArgs la;
      memset(&la, 0, sizeof(la));
      la.ar = ar; la.two_n_free = 2.0 * (double)n_free; la.cap = h->a_cap; la.n_timepoints = Tn; la.keep = keep ? 1 : 0;
      launch_adaptive_persist(h, c, h->stream, (int)std::min<long long>(kLoopAttempts, std::max<long long>(1, max_attempts)), la);
      HIP_OK(hipMemcpyAsync(clk.data(), h->d_clock.p, sizeof(Clock) * B, hipMemcpyDeviceToHost, h->stream));
      HIP_OK(hipStreamSynchronize(h->stream));
      if (*persist_give_up_word(h)) break;
      bool all_done = true;
      long long most = 0, tried = 0;
      rcl = judge(all_done);
      for (size_t m = 0; m < B; ++m) { most = std::max(most, clk[m].accepted); tried = std::max(tried, clk[m].attempts); }
      if (rcl || all_done) break;
      if (tried >= max_attempts) { rcl = out_of_budget(); break; }
      if (keep && most + 4 > h->a_cap) {
        const long long cap = std::min<long long>(cap_fit, std::max<long long>(2 * h->a_cap, most + 4 * kAttemptsPerGraph));
        if (cap < most + 8) { h->err = kNoRoom; rcl = h->isolate_failures ? out_of_budget() : 5; break; }
        if (int rcr = adaptive_room(h, cap, true)) { rcl = h->isolate_failures ? out_of_budget() : rcr; break; }
        h->ck->writer = h;
        c = make_ctx(h);
        refresh_ar();
      }
    }
    if (*persist_give_up_word(h)) {
      // a workgroup was not resident: one launch per stage from now on, and this solve starts over that way -- same process (round-5 advice)
      persist_fell_back(h);
      h->adaptive = false; h->have_traj = false; h->persist_fwd = false;
      return forward_adaptive_impl(h, state0 == rest.data() ? nullptr : state0, timepoints, n_timepoints, rtol, atol, max_attempts, keep, fields, stats);
    }
    HIP_OK(hipEventRecord(h->ev1, h->stream));
    if (rcl) { h->adaptive = false; h->have_traj = false; return rcl; }
    return finish_adaptive(h, clk, keep, fields, stats, true);
  }
  // stage-1 record of the first attempt (accept = 0: nothing to commit)
  hipLaunchKernelGGL(k_prepare, slot_grid(h), dim3(kThreads), 0, h->stream, c, dc, Tn, keep ? 1 : 0);
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  // Same rule as the fixed grid (solve_is_eager): launches that fill the chip are issued eagerly -- instantiating the 256-node graph
  // cost 5-10 ms per call, more than a short solve runs (profiles/r02_adaptive_fixed_cost.txt); small lattices replay a graph,
  // kept in the handle while the arguments baked into it stay the same.
  const long long waves = (long long)pl.batch * ((pl.n_slots + 63) / 64);
  auto prepare_graph = [&]() -> int {
    exec = nullptr;
    if (!(h->use_graph && waves < 2048)) return 0;
    dfx_handle::AdaptiveKey key;
    memset(&key, 0, sizeof(key));
    key.ctx = c; key.n_timepoints = Tn; key.n_partials = n_partials; key.two_n_free = 2.0 * (double)n_free;
    key.keep = keep ? 1 : 0; key.ar = ar;
    if (h->adaptive_exec && memcmp(&key, &h->adaptive_key, sizeof(key)) != 0) {
      (void)hipGraphExecDestroy(h->adaptive_exec); h->adaptive_exec = nullptr;
    }
    if (!h->adaptive_exec) {
      const long long before = h->launches;
      HIP_OK(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
      for (int a = 0; a < kAttemptsPerGraph; ++a) enqueue_attempt();
      hipError_t ce = hipStreamEndCapture(h->stream, &graph);
      h->launches = before;
      if (ce == hipSuccess) ce = hipGraphInstantiate(&h->adaptive_exec, graph, nullptr, nullptr, 0);
      if (graph) (void)hipGraphDestroy(graph);
      if (ce != hipSuccess) {
        h->adaptive_exec = nullptr;
        h->err = std::string("graph capture / instantiate: ") + hipGetErrorString(ce); h->adaptive = false; return 2;
      }
      memcpy(&h->adaptive_key, &key, sizeof(key));
    }
    exec = h->adaptive_exec;
    return 0;
  };
  if (int rcg = prepare_graph()) return rcg;
  long long attempts_issued = 0;
  int rc = 0;
  while (true) {
    if (exec) { HIP_OK(hipGraphLaunch(exec, h->stream)); h->launches += 8LL * kAttemptsPerGraph; }
    else if (h->groups.size() > 1) {
      // eager launches that fill the chip: the member groups advance on their own streams, interleaved launch by launch like the
      // fixed grid (every member carries its own clock, so the groups are independent); the main stream waits for all of them
      // before the clocks are read
      if (int rc2 = fork_groups(h)) return rc2;
      for (int a = 0; a < kAttemptsPerGraph; ++a)
        for (int p = 0; p < 8; ++p)
          for (int gi = 0; gi < (int)h->groups.size(); ++gi) {
            const Group& gr = h->groups[gi];
            launch_phase(p, group_ctx(h, c, gi), gr.stream, slot_grid(h, gr), (unsigned)gr.nm);
          }
      if (int rc2 = join_groups(h)) return rc2;
    }
    else for (int a = 0; a < kAttemptsPerGraph; ++a) enqueue_attempt();
    attempts_issued += kAttemptsPerGraph;
    HIP_OK(hipMemcpyAsync(clk.data(), h->d_clock.p, sizeof(Clock) * B, hipMemcpyDeviceToHost, h->stream));
    HIP_OK(hipStreamSynchronize(h->stream));
    bool all_done = true;
    rc = judge(all_done);
    if (rc || all_done) break;
    if (attempts_issued >= max_attempts) { rc = out_of_budget(); break; }
    if (keep) {       // room for the next round of attempts (every one of them may be accepted)
      long long most = 0;
      for (size_t m = 0; m < B; ++m) most = std::max(most, clk[m].accepted);
      if (most + kAttemptsPerGraph + 4 > h->a_cap) {
        const long long cap = std::min<long long>(cap_fit, std::max<long long>(2 * h->a_cap, most + 4 * kAttemptsPerGraph));
        if (cap < most + kAttemptsPerGraph + 4) { h->err = kNoRoom; rc = h->isolate_failures ? out_of_budget() : 5; break; }
        if (int rcr = adaptive_room(h, cap, true)) { rc = h->isolate_failures ? out_of_budget() : rcr; break; }
        h->ck->writer = h;
        c = make_ctx(h);
        refresh_ar();
        if (int rcg = prepare_graph()) { rc = rcg; break; }
      }
    }
  }
  HIP_OK(hipEventRecord(h->ev1, h->stream));
  if (rc) { h->adaptive = false; h->have_traj = false; return rc; }
  return finish_adaptive(h, clk, keep, fields, stats, false);
}


int dfx_adaptive_step_counts(dfx_handle* h, int32_t* counts) {
  HIP_OK(hipSetDevice(h->device));
  if (!h->have_adaptive_record) { h->err = "adaptive_step_counts: run forward_adaptive first"; return 1; }
  if (h->n_counts > 0) {
    HIP_OK(hipMemcpyAsync(counts, h->d_step_counts.p, sizeof(int) * (size_t)h->pl.batch * h->n_counts, hipMemcpyDeviceToHost, h->stream));
    HIP_OK(hipStreamSynchronize(h->stream));
  }
  return 0;
}

int dfx_adaptive_step_times(dfx_handle* h, int32_t member, double* times, int64_t capacity, int64_t* n) {
  HIP_OK(hipSetDevice(h->device));
  if (!h->have_adaptive_record) { h->err = "adaptive_step_times: run forward_adaptive first"; return 1; }
  if (member < 0 || member >= h->pl.batch) { h->err = "adaptive_step_times: no such member"; return 1; }
  const long long acc = h->accepted_per_member[member];
  if (acc > kAccCap) { h->err = "adaptive_step_times: more than 2^20 accepted steps, times were not recorded"; return 1; }
  *n = acc;
  const long long cnt = std::min<long long>(acc, capacity);
  if (cnt > 0) {
    HIP_OK(hipMemcpyAsync(times, h->d_acc_times.p + (size_t)member * kAccCap, sizeof(double) * cnt, hipMemcpyDeviceToHost, h->stream));
    HIP_OK(hipStreamSynchronize(h->stream));
  }
  return 0;
}
