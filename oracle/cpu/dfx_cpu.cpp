// dfx_cpu.cpp -- C++17 CPU port of the engine (TEST INFRASTRUCTURE + timed CPU baseline).
//
// Exports the same C ABI as libdfx (include/dfx.h) so the same Python binding can drive it, but it
// is built into oracle/cpu/libdfx_cpu.so and is only ever loaded by tests/, smoke() and bench.py's
// cpu_baseline leg.  The product (difflexmm_amd) never loads it.
//
// It runs the algorithm of the reference's hot path -- rhs (difflexmm/dynamics.py:33-55) inside a
// Dormand-Prince step (jax.experimental.ode.runge_kutta_step) -- on host cores, with the force
// hand-derived instead of obtained by autodiff.  The per-ligament formulas are the ones in
// difflexmm_amd/csrc/dfx_physics.h (also compiled into the GPU kernels); their independent check is
// the torch-autograd restatement in oracle/ref_*.py.  Parallelism: OpenMP over blocks.
#define DFX_ABI_LAYOUT_IMPL
#include <math.h>
#include <cmath>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <string>
#include <vector>

#include "../../difflexmm_amd/csrc/dfx_stage.h"
#include "../../difflexmm_amd/csrc/dfx_design.h"

using namespace dfx;

static std::string g_create_error;

struct dfx_handle {
  Plan pl;
  PackedParams pp;
  bool have_params = false;
  std::string err;
  // last forward
  std::vector<double> traj;       // batch * (N+1) * n_blocks * kRec   (when kept)
  std::vector<double> fields;     // batch * T * 2 * n_blocks * 3
  std::vector<double> ts;
  std::vector<int> spis;          // steps per output interval
  std::vector<int64_t> step0;     // first step ordinal of every interval
  std::vector<int32_t> step_counts;  // accepted steps per member and interval (adaptive)
  std::vector<std::vector<double>> acc_times;  // end times of the accepted steps per member (adaptive)
  std::vector<double> t_steps;       // caller-chosen step boundaries (empty: equal steps); per member when ts_stride != 0
  size_t ts_stride = 0, tp_stride = 0;   // elements between members in t_steps / ts (0: one grid for all members)
  int n_tp = 0;                          // output times per member
  bool have_traj = false;
  std::vector<double> view_store[10];  // dfx_kinetic_value_and_grad: arrays behind the returned views
  std::vector<double> zero_state;
  // dfx_forward_adaptive_keep: the accepted steps of every member -- step states (records), step boundaries (t_0 .. t_N), and for every
  // output the step it lies in (-1: before the first step) with its relative position inside that step
  bool adaptive_rec = false;
  std::vector<std::vector<double>> a_traj, a_tsteps, a_theta;
  std::vector<std::vector<int>> a_out_step;
};

static Tables member_tables(const dfx_handle* h, int m) {
  const Plan& pl = h->pl;
  Tables tb;
  tb.n_blocks = pl.n_blocks; tb.n_fns = pl.n_fns; tb.model = pl.model; tb.contact = pl.contact; tb.n_npb = pl.n_npb;
  tb.centroid = h->pp.centroid.data() + (size_t)m * pl.n_blocks * 2;
  tb.slot_info = pl.slot_info.data();
  tb.block_special = pl.block_special.data();
  tb.special = pl.special.data();
  tb.slot_p = h->pp.slot.data() + (size_t)m * pl.n_slots * kSlotParams;
  tb.inv_m = h->pp.inv_m.data() + (size_t)m * pl.n_blocks * 3;
  tb.damping = h->pp.damping.data() + (size_t)m * pl.n_blocks * 3;
  tb.contact_p = h->pp.contact.data() + (size_t)m * 3;
  tb.fns = h->pp.fns.data() + (size_t)m * DFX_MAX_FNS;
  if (pl.n_ovf) {
    tb.ovf_ptr = pl.ovf_ptr.data(); tb.ovf_info = pl.ovf_info.data();
    tb.ovf_p = h->pp.ovf.data() + (size_t)m * pl.n_ovf * kOvfParams;
  }
  return tb;
}

template <int MODEL, int CONTACT>
static void fwd_stage_t(const Tables& tb, const Tableau& T, const FwdStage& st, double* energy) {
  double etot = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : etot)
  for (int b = 0; b < tb.n_blocks; ++b) {
    double f[3] = {0, 0, 0};
    for (int k = 0; k < kSlots; ++k) {
      double fx, fy, fth, e;
      fwd_slot<MODEL, CONTACT>(tb, st.S_in, b * kSlots + k, fx, fy, fth, &e);
      f[0] += fx; f[1] += fy; f[2] += fth;
      etot += e;
    }
    if (st.A)
      for (int d = 0; d < 3; ++d) fwd_dof(tb, T, st, b, d, f[d]);
  }
  if (energy) *energy = etot;
}

static void fwd_stage(const Tables& tb, const Tableau& T, const FwdStage& st, double* energy = nullptr) {
#define CASE(M) case M: if (tb.contact == 2) fwd_stage_t<M, 2>(tb, T, st, energy); else if (tb.contact) fwd_stage_t<M, 1>(tb, T, st, energy); else fwd_stage_t<M, 0>(tb, T, st, energy); break;
  switch (tb.model) { CASE(kNonlinear) CASE(kLinearized) CASE(kSimpleSpring) CASE(kStretchTorsion) }
#undef CASE
}

template <int MODEL, int CONTACT>
static void adj_stage_t(const Tables& tb, const Tableau& T, const AdjStage& st, const GradAcc& acc) {
#pragma omp parallel for schedule(static)
  for (int b = 0; b < tb.n_blocks; ++b) {
    double hsum[3] = {0, 0, 0};
    for (int k = 0; k < kSlots; ++k) {
      double hx, hy, hth;
      adj_slot<MODEL, CONTACT>(tb, st.S, st.W, b * kSlots + k, acc, hx, hy, hth);
      hsum[0] += hx; hsum[1] += hy; hsum[2] += hth;
    }
    for (int d = 0; d < 3; ++d) adj_dof(tb, T, st, acc, b, d, hsum[d]);
  }
}

static void adj_stage(const Tables& tb, const Tableau& T, const AdjStage& st, const GradAcc& acc) {
#define CASE(M) case M: if (tb.contact == 2) adj_stage_t<M, 2>(tb, T, st, acc); else if (tb.contact) adj_stage_t<M, 1>(tb, T, st, acc); else adj_stage_t<M, 0>(tb, T, st, acc); break;
  switch (tb.model) { CASE(kNonlinear) CASE(kLinearized) CASE(kSimpleSpring) CASE(kStretchTorsion) }
#undef CASE
}

static void snapshot(const double* S, int nb, double* out /* (2, nb, 3) */) {
  for (int b = 0; b < nb; ++b)
    for (int d = 0; d < 3; ++d) {
      out[b * 3 + d] = S[(size_t)b * kRec + d];
      out[nb * 3 + b * 3 + d] = S[(size_t)b * kRec + 5 + d];
    }
}

extern "C" {

int dfx_create(const dfx_problem* problem, dfx_handle** out) {
  dfx_handle* h = new dfx_handle();
  if (build_plan(problem, h->pl, h->err)) { g_create_error = h->err; delete h; return 1; }
  *out = h;
  return 0;
}

int dfx_destroy(dfx_handle* h) { delete h; return 0; }

const char* dfx_last_error(const dfx_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int dfx_set_params(dfx_handle* h, const dfx_params* params) {
  if (pack_params(h->pl, params, h->pp, h->err, false)) return 1;
  h->have_params = true;
  h->have_traj = false;
  return 0;
}

int dfx_reserve(dfx_handle*, int64_t, int32_t, int32_t) { return 0; }
int dfx_share_checkpoint(dfx_handle*, dfx_handle*) { return 0; }      // host memory: every handle keeps its own trajectory

int dfx_forward_grid(dfx_handle* h, const double* state0, const double* timepoints, int32_t n_timepoints,
                     const int32_t* steps_per_interval, const double* step_times, int32_t keep_trajectory,
                     double* fields, dfx_stats* stats);

int dfx_forward(dfx_handle* h, const double* state0, const double* timepoints, int32_t n_timepoints,
                int32_t steps_per_interval, int32_t keep_trajectory, double* fields, dfx_stats* stats) {
  if (n_timepoints < 1 || steps_per_interval < 1) { h->err = "forward: need >= 1 timepoint and >= 1 step per interval"; return 1; }
  std::vector<int32_t> spis((size_t)std::max(0, n_timepoints - 1), steps_per_interval);
  return dfx_forward_grid(h, state0, timepoints, n_timepoints, spis.data(), nullptr, keep_trajectory, fields, stats);
}

static int forward_grid_impl(dfx_handle* h, const double* state0, const double* timepoints, int32_t n_timepoints,
                             const int32_t* steps_per_interval, const double* step_times, int32_t keep_trajectory,
                             double* fields, dfx_stats* stats, bool per_member);

int dfx_forward_grid(dfx_handle* h, const double* state0, const double* timepoints, int32_t n_timepoints,
                     const int32_t* steps_per_interval, const double* step_times, int32_t keep_trajectory,
                     double* fields, dfx_stats* stats) {
  return forward_grid_impl(h, state0, timepoints, n_timepoints, steps_per_interval, step_times, keep_trajectory, fields, stats, false);
}

int dfx_forward_grid_members(dfx_handle* h, const double* state0, const double* timepoints, int32_t n_timepoints,
                             const int32_t* steps_per_interval, const double* step_times, int32_t keep_trajectory,
                             double* fields, dfx_stats* stats) {
  if (!step_times) { h->err = "forward_grid_members: step_times (batch, n_steps + 1) required"; return 1; }
  return forward_grid_impl(h, state0, timepoints, n_timepoints, steps_per_interval, step_times, keep_trajectory, fields, stats, true);
}

static int forward_grid_impl(dfx_handle* h, const double* state0, const double* timepoints, int32_t n_timepoints,
                             const int32_t* steps_per_interval, const double* step_times, int32_t keep_trajectory,
                             double* fields, dfx_stats* stats, bool per_member) {
  if (!h->have_params) { h->err = "forward: set_params first"; return 1; }
  if (n_timepoints < 1) { h->err = "forward: need >= 1 timepoint and >= 1 step per interval"; return 1; }
  const Plan& pl = h->pl;
  const Tableau& T = pl.tab;
  const int nb = pl.n_blocks, B = pl.batch, Tn = n_timepoints;
  const size_t rec = (size_t)nb * kRec;
  h->spis.assign(steps_per_interval, steps_per_interval + (Tn - 1));
  h->step0.assign(Tn, 0);
  for (int k = 0; k + 1 < Tn; ++k) {
    if (h->spis[k] < 1) { h->err = "forward: need >= 1 timepoint and >= 1 step per interval"; return 1; }
    h->step0[k + 1] = h->step0[k] + h->spis[k];
  }
  const int64_t N = h->step0[Tn - 1];
  h->t_steps.clear();
  h->ts_stride = per_member ? (size_t)N + 1 : 0;
  h->tp_stride = per_member ? (size_t)Tn : 0;
  const int n_grids = per_member ? B : 1;
  if (step_times) {
    h->t_steps.assign(step_times, step_times + (size_t)n_grids * (N + 1));
    for (int g = 0; g < n_grids; ++g) {
      const double* tsg = h->t_steps.data() + (size_t)g * (N + 1);
      for (int64_t n = 0; n < N; ++n)
        if (!(tsg[n + 1] > tsg[n])) { h->err = "forward: step_times must be strictly increasing"; return 1; }
      for (int k = 0; k < Tn; ++k)
        if (tsg[h->step0[k]] != timepoints[(size_t)g * Tn + k]) { h->err = "forward: step_times must contain every timepoint at the start of its interval"; return 1; }
    }
  }
  if (!state0) { h->zero_state.assign((size_t)B * nb * 6, 0.0); state0 = h->zero_state.data(); }   // NULL: at rest
  const bool own_grid = !h->t_steps.empty();
  auto t_begin = std::chrono::steady_clock::now();
  h->ts.assign(timepoints, timepoints + (size_t)n_grids * Tn);
  h->n_tp = Tn;
  const double* timepoints_all = timepoints;
  h->have_traj = keep_trajectory != 0;
  h->adaptive_rec = false;
  h->fields.assign((size_t)B * Tn * nb * 6, 0.0);
  if (keep_trajectory) h->traj.assign((size_t)B * (N + 1) * rec, 0.0);
  std::vector<double> Ybuf(2 * rec), Sbuf(2 * rec), A((size_t)T.s * nb * 3);
  for (int m = 0; m < B; ++m) {
    Tables tb = member_tables(h, m);
    timepoints = timepoints_all + (size_t)m * h->tp_stride;            // this member's output times and step boundaries
    const double* t_steps = own_grid ? h->t_steps.data() + (size_t)m * h->ts_stride : nullptr;
    double* fm = h->fields.data() + (size_t)m * Tn * nb * 6;
    double* tr = keep_trajectory ? h->traj.data() + (size_t)m * (N + 1) * rec : nullptr;
    double* Y = tr ? tr : Ybuf.data();
    for (int b = 0; b < nb; ++b)
      for (int d = 0; d < 3; ++d) init_dof(tb, state0 + (size_t)m * nb * 6, timepoints[0], Y, b, d);
    snapshot(Y, nb, fm);
    int64_t n = 0;
    for (int k = 0; k + 1 < Tn; ++k) {
      const int spi = h->spis[k];
      const double heq = (timepoints[k + 1] - timepoints[k]) / spi;
      for (int j = 0; j < spi; ++j, ++n) {
        const double t = own_grid ? t_steps[n] : timepoints[k] + j * heq;
        const double hh = own_grid ? t_steps[n + 1] - t : heq;
        double* Ynext = tr ? tr + (size_t)(n + 1) * rec : (Y == Ybuf.data() ? Ybuf.data() + rec : Ybuf.data());
        for (int i = 0; i < T.s; ++i) {
          FwdStage st;
          st.S_in = i == 0 ? Y : Sbuf.data() + (size_t)(i & 1) * rec;
          st.S_out = i == T.s - 1 ? Ynext : Sbuf.data() + (size_t)((i + 1) & 1) * rec;
          st.Y = Y; st.A = A.data(); st.i = i; st.h = hh;
          st.t_i = t + T.c[i] * hh; st.t_next = t + T.c[i + 1] * hh;
          fwd_stage(tb, T, st);
        }
        Y = Ynext;
      }
      snapshot(Y, nb, fm + (size_t)(k + 1) * nb * 6);
    }
  }
  if (fields) memcpy(fields, h->fields.data(), sizeof(double) * h->fields.size());
  if (stats) {
    memset(stats, 0, sizeof(*stats));
    stats->steps = N; stats->rhs_evals = N * T.s; stats->launches = 0;
    stats->kernel_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
  }
  return 0;
}

}  // extern "C" (reopened below)

// ---- adaptive Dormand-Prince with the semantics of jax.experimental.ode.odeint (see oracle/ref_ode.py) -----------
namespace {
struct AdaptiveMember {
  const Tables& tb;
  const Plan& pl;
  std::vector<double> S, Abuf;
  std::vector<char> con;   // constrained flag per DOF
  AdaptiveMember(const Tables& t, const Plan& p) : tb(t), pl(p), S((size_t)p.n_blocks * kRec), Abuf((size_t)p.tab.s * p.n_blocks * 3), con((size_t)p.n_blocks * 3, 0) {
    for (int b = 0; b < p.n_blocks; ++b) {
      int sidx = p.block_special[b];
      if (sidx >= 0) for (int d = 0; d < 3; ++d) con[b * 3 + d] = (p.special[sidx].con_mask >> d) & 1;
    }
  }
  // k = f(y, t): kq = v (0 on constrained DOFs), kv = acceleration
  void rhs(const double* q, const double* v, double t, double* kq, double* kv) {
    const int nb = pl.n_blocks, nd = nb * 3;
    std::vector<double> y(2 * (size_t)nd);
    memcpy(y.data(), q, sizeof(double) * nd);
    memcpy(y.data() + nd, v, sizeof(double) * nd);
    for (int b = 0; b < nb; ++b)
      for (int d = 0; d < 3; ++d) init_dof(tb, y.data(), t, S.data(), b, d);
    FwdStage st;
    st.S_in = S.data(); st.S_out = nullptr; st.Y = S.data(); st.A = Abuf.data(); st.i = 0; st.h = 0.0; st.t_i = t; st.t_next = t;
    fwd_stage(tb, pl.tab, st);
    for (int i = 0; i < nd; ++i) { kq[i] = con[i] ? 0.0 : v[i]; kv[i] = Abuf[i]; }
  }
};
}  // namespace

static int forward_adaptive_impl(dfx_handle* h, const double* state0, const double* timepoints, int32_t n_timepoints,
                                 double rtol, double atol, int64_t max_attempts, int keep, double* fields, dfx_stats* stats) {
  if (!h->have_params) { h->err = "forward_adaptive: set_params first"; return 1; }
  if (n_timepoints < 1) { h->err = "forward_adaptive: need >= 1 timepoint"; return 1; }
  const Plan& pl = h->pl;
  const int nb = pl.n_blocks, nd = nb * 3, B = pl.batch, Tn = n_timepoints;
  const Dopri D = make_dopri();
  if (!state0) { h->zero_state.assign((size_t)B * nb * 6, 0.0); state0 = h->zero_state.data(); }   // NULL: at rest
  h->ts.assign(timepoints, timepoints + Tn);
  h->n_tp = Tn; h->tp_stride = 0; h->ts_stride = 0;
  h->have_traj = false;
  h->adaptive_rec = false;
  h->fields.assign((size_t)B * Tn * nb * 6, 0.0);
  h->step_counts.assign((size_t)B * std::max(0, Tn - 1), 0);
  h->acc_times.assign(B, {});
  h->a_traj.assign(keep ? B : 0, {}); h->a_tsteps.assign(keep ? B : 0, {}); h->a_theta.assign(keep ? B : 0, {}); h->a_out_step.assign(keep ? B : 0, {});
  const auto t_begin = std::chrono::steady_clock::now();
  int64_t max_acc = 0, max_try = 0;
  for (int m = 0; m < B; ++m) {
    Tables tb = member_tables(h, m);
    AdaptiveMember M(tb, pl);
    std::vector<double> q(state0 + (size_t)m * 2 * nd, state0 + (size_t)m * 2 * nd + nd), v(state0 + (size_t)m * 2 * nd + nd, state0 + (size_t)(m + 1) * 2 * nd);
    std::vector<double> kq(7 * (size_t)nd), kv(7 * (size_t)nd), yq(nd), yv(nd), q1(nd), v1(nd), qm(nd), vm(nd);
    int n_free = 0;
    for (int i = 0; i < nd; ++i) n_free += !M.con[i];
    double* fm = h->fields.data() + (size_t)m * Tn * nb * 6;
    auto write_out = [&](int k, const double* oq, const double* ov) {
      // constrained DOFs follow c(t_k), c'(t_k) exactly (dynamics.py:132-134)
      std::vector<double> y(2 * (size_t)nd), S((size_t)nb * kRec);
      memcpy(y.data(), oq, sizeof(double) * nd); memcpy(y.data() + nd, ov, sizeof(double) * nd);
      for (int b = 0; b < nb; ++b) for (int d = 0; d < 3; ++d) init_dof(tb, y.data(), timepoints[k], S.data(), b, d);
      snapshot(S.data(), nb, fm + (size_t)k * nb * 6);
    };
    double t = timepoints[0];
    M.rhs(q.data(), v.data(), t, kq.data(), kv.data());
    write_out(0, q.data(), v.data());
    auto keep_state = [&]() {       // the record of the current state (constrained DOFs at c(t)): what the reverse sweep restarts a step from
      std::vector<double> y(2 * (size_t)nd), S((size_t)nb * kRec);
      memcpy(y.data(), q.data(), sizeof(double) * nd); memcpy(y.data() + nd, v.data(), sizeof(double) * nd);
      for (int b = 0; b < nb; ++b) for (int d = 0; d < 3; ++d) init_dof(tb, y.data(), t, S.data(), b, d);
      h->a_traj[m].insert(h->a_traj[m].end(), S.begin(), S.end());
      h->a_tsteps[m].push_back(t);
    };
    if (keep) { keep_state(); h->a_theta[m].assign(Tn, 0.0); h->a_out_step[m].assign(Tn, -1); }
    // initial step (Hairer II.4 as restated by jax, order 4)
    double d0 = 0, d1 = 0, d2 = 0;
    for (int i = 0; i < nd; ++i) if (!M.con[i]) {
      double sq = atol + fabs(q[i]) * rtol, sv = atol + fabs(v[i]) * rtol;
      d0 += (q[i] / sq) * (q[i] / sq) + (v[i] / sv) * (v[i] / sv);
      d1 += (kq[i] / sq) * (kq[i] / sq) + (kv[i] / sv) * (kv[i] / sv);
    }
    d0 = sqrt(d0); d1 = sqrt(d1);
    double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
    for (int i = 0; i < nd; ++i) { yq[i] = q[i] + h0 * kq[i]; yv[i] = v[i] + h0 * kv[i]; }
    M.rhs(yq.data(), yv.data(), t + h0, kq.data() + nd, kv.data() + nd);
    for (int i = 0; i < nd; ++i) if (!M.con[i]) {
      double sq = atol + fabs(q[i]) * rtol, sv = atol + fabs(v[i]) * rtol;
      double a = (kq[nd + i] - kq[i]) / sq, b = (kv[nd + i] - kv[i]) / sv;
      d2 += a * a + b * b;
    }
    d2 = sqrt(d2) / h0;
    double h1 = (d1 <= 1e-15 && d2 <= 1e-15) ? std::max(1e-6, h0 * 1e-3) : pow(0.01 / (d1 + d2), 1.0 / 5.0);
    double dt = std::min(100.0 * h0, h1);
    int64_t n_try = 0, n_acc = 0;
    double t_last = t;
    std::vector<double> c_q0, c_v0;  // state at the start of the last accepted step (for dense output)
    bool have_step = false;
    double h_acc = 0.0;
    std::vector<double> sq0(nd), sv0(nd), sq1(nd), sv1(nd), smq(nd), smv(nd), f0q(nd), f0v(nd), f1q(nd), f1v(nd);
    for (int k = 1; k < Tn; ++k) {
      const double target = timepoints[k];
      while (t < target && dt > 0) {
        if (n_try >= max_attempts) { h->err = "forward_adaptive: step budget exceeded"; return 4; }
        for (int i = 1; i < 7; ++i) {
          for (int x = 0; x < nd; ++x) {
            double aq = 0, av = 0;
            for (int j = 0; j < i; ++j) { aq += D.a[i][j] * kq[(size_t)j * nd + x]; av += D.a[i][j] * kv[(size_t)j * nd + x]; }
            yq[x] = q[x] + dt * aq; yv[x] = v[x] + dt * av;
          }
          M.rhs(yq.data(), yv.data(), t + D.c[i] * dt, kq.data() + (size_t)i * nd, kv.data() + (size_t)i * nd);
        }
        // y1 = stage-6 state (a[6] = solution weights); error estimate
        double r2 = 0.0;
        for (int x = 0; x < nd; ++x) {
          q1[x] = yq[x]; v1[x] = yv[x];
          if (M.con[x]) continue;
          double eq = 0, ev = 0;
          for (int j = 0; j < 7; ++j) { eq += D.e[j] * kq[(size_t)j * nd + x]; ev += D.e[j] * kv[(size_t)j * nd + x]; }
          eq *= dt; ev *= dt;
          double tq = atol + rtol * std::max(fabs(q[x]), fabs(q1[x])), tv = atol + rtol * std::max(fabs(v[x]), fabs(v1[x]));
          r2 += (eq / tq) * (eq / tq) + (ev / tv) * (ev / tv);
        }
        const double ratio = sqrt(r2 / (2.0 * n_free));
        if (!(ratio == ratio)) { h->err = "forward_adaptive: non-finite error estimate"; return 3; }
        ++n_try;
        const double dt_new = dopri_next_step(dt, ratio);
        if (ratio <= 1.0) {
          for (int x = 0; x < nd; ++x) {
            double mq = 0, mv = 0;
            for (int j = 0; j < 7; ++j) { mq += D.cm[j] * kq[(size_t)j * nd + x]; mv += D.cm[j] * kv[(size_t)j * nd + x]; }
            sq0[x] = q[x]; sv0[x] = v[x]; sq1[x] = q1[x]; sv1[x] = v1[x];
            smq[x] = q[x] + dt * mq; smv[x] = v[x] + dt * mv;
            f0q[x] = kq[x]; f0v[x] = kv[x]; f1q[x] = kq[(size_t)6 * nd + x]; f1v[x] = kv[(size_t)6 * nd + x];
            q[x] = q1[x]; v[x] = v1[x];
            kq[x] = f1q[x]; kv[x] = f1v[x];   // FSAL
          }
          t_last = t; h_acc = dt; t += dt; have_step = true;
          ++n_acc;
          ++h->step_counts[(size_t)m * (Tn - 1) + (k - 1)];   // the interval that contains the start of the step
          h->acc_times[m].push_back(t);
          if (keep) keep_state();
        }
        dt = dt_new;
      }
      if (have_step) {
        const double r = (target - t_last) / (t - t_last);
        for (int x = 0; x < nd; ++x) {
          yq[x] = dopri_dense(sq0[x], sq1[x], smq[x], f0q[x], f1q[x], h_acc, r);
          yv[x] = dopri_dense(sv0[x], sv1[x], smv[x], f0v[x], f1v[x], h_acc, r);
        }
        write_out(k, yq.data(), yv.data());
        if (keep) { h->a_theta[m][k] = r; h->a_out_step[m][k] = (int)n_acc - 1; }
      } else {
        write_out(k, q.data(), v.data());
      }
    }
    max_acc = std::max(max_acc, n_acc); max_try = std::max(max_try, n_try);
  }
  if (fields) memcpy(fields, h->fields.data(), sizeof(double) * h->fields.size());
  if (stats) {
    memset(stats, 0, sizeof(*stats)); stats->steps = max_acc; stats->rhs_evals = 6 * max_try + 2;
    stats->kernel_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    stats->checkpoint_records = keep ? 1 : 0;
  }
  if (keep) { h->adaptive_rec = true; h->have_traj = true; }
  return 0;
}

extern "C" int dfx_forward_adaptive(dfx_handle* h, const double* state0, const double* timepoints, int32_t n_timepoints,
                                    double rtol, double atol, int64_t max_attempts, double* fields, dfx_stats* stats) {
  return forward_adaptive_impl(h, state0, timepoints, n_timepoints, rtol, atol, max_attempts, 0, fields, stats);
}
extern "C" int dfx_forward_adaptive_keep(dfx_handle* h, const double* state0, const double* timepoints, int32_t n_timepoints,
                                         double rtol, double atol, int64_t max_attempts, int32_t keep_trajectory, double* fields, dfx_stats* stats) {
  return forward_adaptive_impl(h, state0, timepoints, n_timepoints, rtol, atol, max_attempts, keep_trajectory != 0, fields, stats);
}

extern "C" {

// Reverse sweep of an adaptive solve that kept its accepted steps (dfx_forward_adaptive_keep): the exact discrete adjoint of those steps with
// the step sizes frozen -- the outputs are interpolated INSIDE steps (jax's quartic dense output), so their cotangents enter through the
// stage slopes: an output at relative position r of step n adds g to lambda_n and h_n B_j(r) g to Kbar_j (dopri_dense_weights); the
// seventh slope k_6 = f(y_{n+1}) is the first slope of step n + 1 (FSAL), so its share joins Kbar_0 of that step -- or, after the last
// step, one extra Hessian-vector product at the final state.
static int run_adjoint_dense(dfx_handle* h, const std::vector<double>& Gall, dfx_grads* grads, dfx_stats* stats) {
  const Plan& pl = h->pl;
  const Tableau& T = pl.tab;
  const Dopri D = make_dopri();
  const int nb = pl.n_blocks, B = pl.batch, Tn = h->n_tp;
  const size_t rec = (size_t)nb * kRec;
  if (T.s != 6) { h->err = "adjoint: the adaptive solve is defined for the dopri5 tableau"; return 1; }
  auto t_begin = std::chrono::steady_clock::now();
  const int nsp = pl.n_special > 0 ? pl.n_special : 1;
  std::vector<double> slot_g((size_t)B * pl.n_slots * kSlotGrads, 0.0), blk_g((size_t)B * nb * 6, 0.0),
      fn_g((size_t)B * nsp * DFX_MAX_FNS * DFX_FN_PARAMS, 0.0), cen_g((size_t)B * nb * 2, 0.0),
      ovf_g((size_t)B * pl.n_ovf * kOvfGrads + 1, 0.0);
  std::vector<double> Sst((size_t)T.s * rec), A((size_t)T.s * nb * 3), YB((size_t)T.s * nb * 6), LAM((size_t)nb * 6),
      W(2 * (size_t)nb * 3), KQ(2 * (size_t)nb * 3), src((size_t)nb * 6), gsum((size_t)nb * 6);
  int64_t max_steps = 0;
  for (int m = 0; m < B; ++m) {
    Tables tb = member_tables(h, m);
    GradAcc acc{slot_g.data() + (size_t)m * pl.n_slots * kSlotGrads, blk_g.data() + (size_t)m * nb * 6,
                fn_g.data() + (size_t)m * nsp * DFX_MAX_FNS * DFX_FN_PARAMS, cen_g.data() + (size_t)m * nb * 2,
                ovf_g.data() + (size_t)m * pl.n_ovf * kOvfGrads};
    const std::vector<double>& tst = h->a_tsteps[m];
    const int64_t N = (int64_t)tst.size() - 1;
    max_steps = std::max(max_steps, N);
    const double* tr = h->a_traj[m].data();
    const double* G = Gall.data() + (size_t)m * Tn * nb * 6;
    // outputs of every step, with their slope weights
    std::vector<std::vector<int>> outs((size_t)std::max<int64_t>(N, 0) + 1);
    std::vector<double> Bw((size_t)Tn * 7, 0.0);
    for (int k = 1; k < Tn; ++k) {
      const int n = h->a_out_step[m][k];
      if (n >= 0) { outs[n].push_back(k); dopri_dense_weights(h->a_theta[m][k], D.a[6], D.cm, Bw.data() + (size_t)k * 7); }
    }
    // sum over the outputs of step n of  scale * B_j g  (j < 0: of g itself), into `dst`
    auto add_outputs = [&](int64_t n, int j, double scale, std::vector<double>& dst) {
      if (n < 0 || n >= N) return;
      for (int k : outs[n]) {
        const double w = j < 0 ? scale : scale * Bw[(size_t)k * 7 + j];
        const double* g = G + (size_t)k * nb * 6;
        for (size_t x = 0; x < (size_t)nb * 6; ++x) dst[x] += w * g[x];
      }
    };
    auto hstep = [&](int64_t n) { return (n >= 0 && n < N) ? tst[n + 1] - tst[n] : 0.0; };
    int cur = 0;
    std::fill(LAM.begin(), LAM.end(), 0.0);
    std::fill(YB.begin(), YB.end(), 0.0);
    // Kbar of the extra evaluation at the final state: h_{N-1} B_6 g of the outputs inside the last step
    std::fill(src.begin(), src.end(), 0.0);
    add_outputs(N - 1, 6, hstep(N - 1), src);
    for (int b = 0; b < nb; ++b)
      for (int d = 0; d < 3; ++d) {
        const int sidx = tb.block_special[b];
        const bool con = sidx >= 0 && ((tb.special[sidx].con_mask >> d) & 1);
        KQ[(size_t)cur * nb * 3 + b * 3 + d] = con ? 0.0 : src[b * 6 + d];
        W[(size_t)cur * nb * 3 + b * 3 + d] = con ? 0.0 : src[b * 6 + 3 + d] * tb.inv_m[b * 3 + d];
      }
    for (int64_t n = N; n >= 0; --n) {
      const double t = tst[n], hh = hstep(n);
      const double* Y = tr + (size_t)n * rec;
      const int i_hi = n == N ? 0 : T.s - 1;            // n == N: only the evaluation at the final state (a step of size zero)
      for (int i = 0; i <= i_hi; ++i) {       // stage records and accelerations of step n, recomputed from its state
        FwdStage st;
        st.S_in = i == 0 ? Y : Sst.data() + (size_t)i * rec;
        st.S_out = i == i_hi ? nullptr : Sst.data() + (size_t)(i + 1) * rec;
        st.Y = Y; st.A = A.data(); st.i = i; st.h = hh;
        st.t_i = t + T.c[i] * hh; st.t_next = t + T.c[i + 1] * hh;
        fwd_stage(tb, T, st);
      }
      for (int i = i_hi; i >= 0; --i) {
        std::fill(src.begin(), src.end(), 0.0);
        const double* Gadd = nullptr;
        if (i > 0) {
          add_outputs(n, i - 1, hh, src);                                  // Kbar_{i-1} of this step
          if (i == 1) add_outputs(n - 1, 6, hstep(n - 1), src);            // ... and the FSAL slope of the previous one
        } else {
          add_outputs(n - 1, T.s - 1, hstep(n - 1), src);                  // Kbar_{s-1} of the previous step
          std::fill(gsum.begin(), gsum.end(), 0.0);
          add_outputs(n, -1, 1.0, gsum);                                   // lambda_n += g of the outputs inside this step
          if (n == 0) for (size_t x = 0; x < (size_t)nb * 6; ++x) gsum[x] += G[x];      // ... and of the initial state (row 0) and of
          for (int k = 1; k < Tn; ++k)                                                  // outputs produced before any step
            if (n == 0 && h->a_out_step[m][k] < 0) for (size_t x = 0; x < (size_t)nb * 6; ++x) gsum[x] += G[(size_t)k * nb * 6 + x];
          Gadd = gsum.data();
        }
        AdjStage st;
        st.S = i == 0 ? Y : Sst.data() + (size_t)i * rec;
        st.A = A.data();
        st.W = W.data() + (size_t)cur * nb * 3; st.KQ = KQ.data() + (size_t)cur * nb * 3;
        st.W_out = W.data() + (size_t)(1 - cur) * nb * 3; st.KQ_out = KQ.data() + (size_t)(1 - cur) * nb * 3;
        st.YB = YB.data(); st.LAM = LAM.data();
        st.G = Gadd; st.src = src.data();
        st.i = i; st.local_only = 0; st.t_i = t + T.c[i] * hh; st.h = hh; st.h_prev = hstep(n - 1);
        adj_stage(tb, T, st, acc);
        cur = 1 - cur;
      }
      if (n == N) std::fill(YB.begin(), YB.end(), 0.0);      // (the zero-size step's Ybar_0 has been summed into lambda_N)
    }
    if (grads && grads->state0)
      for (int b = 0; b < nb; ++b)
        for (int d = 0; d < 3; ++d) {
          grads->state0[(size_t)m * nb * 6 + b * 3 + d] = LAM[b * 6 + d];
          grads->state0[(size_t)m * nb * 6 + nb * 3 + b * 3 + d] = LAM[b * 6 + 3 + d];
        }
  }
  if (grads) unpack_grads(pl, slot_g, blk_g, fn_g, h->pp.inv_m, grads, &ovf_g);
  if (grads && grads->block_centroids) memcpy(grads->block_centroids, cen_g.data(), sizeof(double) * cen_g.size());
  if (stats) {
    memset(stats, 0, sizeof(*stats));
    stats->steps = max_steps; stats->rhs_evals = max_steps * T.s; stats->checkpoint_records = 1;
    stats->kernel_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
  }
  return 0;
}

static int run_adjoint(dfx_handle* h, const std::vector<double>& Gall /* batch*T*nb*6 [q(3) v(3)] per block */,
                       dfx_grads* grads, dfx_stats* stats) {
  if (!h->have_traj) { h->err = "adjoint: run forward with keep_trajectory=1 first"; return 1; }
  if (h->adaptive_rec) return run_adjoint_dense(h, Gall, grads, stats);
  const Plan& pl = h->pl;
  const Tableau& T = pl.tab;
  const int nb = pl.n_blocks, B = pl.batch, Tn = h->n_tp;
  const size_t rec = (size_t)nb * kRec;
  const int64_t N = h->step0[Tn - 1];
  auto t_begin = std::chrono::steady_clock::now();
  const int nsp = pl.n_special > 0 ? pl.n_special : 1;
  std::vector<double> slot_g((size_t)B * pl.n_slots * kSlotGrads, 0.0), blk_g((size_t)B * nb * 6, 0.0),
      fn_g((size_t)B * nsp * DFX_MAX_FNS * DFX_FN_PARAMS, 0.0), cen_g((size_t)B * nb * 2, 0.0),
      ovf_g((size_t)B * pl.n_ovf * kOvfGrads + 1, 0.0);
  std::vector<double> Sst((size_t)T.s * rec), A((size_t)T.s * nb * 3), YB((size_t)T.s * nb * 6), LAM((size_t)nb * 6),
      W(2 * (size_t)nb * 3), KQ(2 * (size_t)nb * 3);
  for (int m = 0; m < B; ++m) {
    Tables tb = member_tables(h, m);
    GradAcc acc{slot_g.data() + (size_t)m * pl.n_slots * kSlotGrads, blk_g.data() + (size_t)m * nb * 6,
                fn_g.data() + (size_t)m * nsp * DFX_MAX_FNS * DFX_FN_PARAMS, cen_g.data() + (size_t)m * nb * 2,
                ovf_g.data() + (size_t)m * pl.n_ovf * kOvfGrads};
    const double* tr = h->traj.data() + (size_t)m * (N + 1) * rec;
    const double* G = Gall.data() + (size_t)m * Tn * nb * 6;
    int cur = 0;
    const bool own_grid = !h->t_steps.empty();
    const double* t_steps = own_grid ? h->t_steps.data() + (size_t)m * h->ts_stride : nullptr;     // this member's grid
    const double* ts_m = h->ts.data() + (size_t)m * h->tp_stride;
    const double h_last = Tn > 1 ? (own_grid ? t_steps[N] - t_steps[N - 1] : (ts_m[Tn - 1] - ts_m[Tn - 2]) / h->spis[Tn - 2]) : 0.0;
    for (int b = 0; b < nb; ++b)
      for (int d = 0; d < 3; ++d)
        adj_begin_dof(tb, T, G + (size_t)(Tn - 1) * nb * 6, h_last, LAM.data(), W.data() + (size_t)cur * nb * 3, KQ.data() + (size_t)cur * nb * 3, b, d);
    for (int k = Tn - 2; k >= 0; --k) {
      const int spi = h->spis[k];
      const double heq = (ts_m[k + 1] - ts_m[k]) / spi;
      for (int j = spi - 1; j >= 0; --j) {
        const int64_t n = h->step0[k] + j;
        const double t = own_grid ? t_steps[n] : ts_m[k] + j * heq;
        const double hh = own_grid ? t_steps[n + 1] - t : heq;
        const double* Y = tr + (size_t)n * rec;
        // recompute the stage records of step n
        for (int i = 0; i < T.s; ++i) {
          FwdStage st;
          st.S_in = i == 0 ? Y : Sst.data() + (size_t)i * rec;
          st.S_out = i == T.s - 1 ? nullptr : Sst.data() + (size_t)(i + 1) * rec;
          st.Y = Y; st.A = A.data(); st.i = i; st.h = hh;
          st.t_i = t + T.c[i] * hh; st.t_next = t + T.c[i + 1] * hh;
          fwd_stage(tb, T, st);
        }
        for (int i = T.s - 1; i >= 0; --i) {
          AdjStage st;
          st.S = i == 0 ? Y : Sst.data() + (size_t)i * rec;
          st.A = A.data();
          st.W = W.data() + (size_t)cur * nb * 3; st.KQ = KQ.data() + (size_t)cur * nb * 3;
          st.W_out = W.data() + (size_t)(1 - cur) * nb * 3; st.KQ_out = KQ.data() + (size_t)(1 - cur) * nb * 3;
          st.YB = YB.data(); st.LAM = LAM.data();
          st.G = (i == 0 && j == 0) ? G + (size_t)k * nb * 6 : nullptr;
          st.i = i; st.local_only = 0; st.t_i = t + T.c[i] * hh; st.h = hh;
          st.h_prev = own_grid ? (n > 0 ? t - t_steps[n - 1] : 0.0)
                               : (j > 0 ? hh : (k > 0 ? (ts_m[k] - ts_m[k - 1]) / h->spis[k - 1] : 0.0));
          adj_stage(tb, T, st, acc);
          cur = 1 - cur;
        }
      }
    }
    if (Tn == 1) {
      // no steps: lambda_0 = G_0 (already placed by adj_begin_dof)
    }
    if (grads && grads->state0)
      for (int b = 0; b < nb; ++b)
        for (int d = 0; d < 3; ++d) {
          grads->state0[(size_t)m * nb * 6 + b * 3 + d] = LAM[b * 6 + d];
          grads->state0[(size_t)m * nb * 6 + nb * 3 + b * 3 + d] = LAM[b * 6 + 3 + d];
        }
  }
  if (grads) unpack_grads(pl, slot_g, blk_g, fn_g, h->pp.inv_m, grads, &ovf_g);
  if (grads && grads->block_centroids) memcpy(grads->block_centroids, cen_g.data(), sizeof(double) * cen_g.size());
  if (stats) {
    memset(stats, 0, sizeof(*stats));
    stats->steps = N; stats->rhs_evals = N * T.s;
    stats->kernel_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
  }
  return 0;
}

int dfx_adaptive_step_counts(dfx_handle* h, int32_t* counts) {
  if (h->step_counts.empty() && h->n_tp > 1) { h->err = "adaptive_step_counts: run forward_adaptive first"; return 1; }
  if (!h->step_counts.empty()) memcpy(counts, h->step_counts.data(), sizeof(int32_t) * h->step_counts.size());
  return 0;
}

int dfx_adaptive_step_times(dfx_handle* h, int32_t member, double* times, int64_t capacity, int64_t* n) {
  if (member < 0 || member >= (int)h->acc_times.size()) { h->err = "adaptive_step_times: run forward_adaptive first"; return 1; }
  const std::vector<double>& a = h->acc_times[member];
  *n = (int64_t)a.size();
  for (int64_t i = 0; i < std::min<int64_t>(capacity, (int64_t)a.size()); ++i) times[i] = a[i];
  return 0;
}

int dfx_adjoint(dfx_handle* h, const double* fields_bar, dfx_grads* grads, dfx_stats* stats) {
  const Plan& pl = h->pl;
  const int nb = pl.n_blocks, B = pl.batch, Tn = h->n_tp;
  std::vector<double> G((size_t)B * Tn * nb * 6);
  for (int m = 0; m < B; ++m)
    for (int k = 0; k < Tn; ++k) {
      const double* fb = fields_bar + ((size_t)m * Tn + k) * nb * 6;
      double* g = G.data() + ((size_t)m * Tn + k) * nb * 6;
      for (int b = 0; b < nb; ++b)
        for (int d = 0; d < 3; ++d) { g[b * 6 + d] = fb[b * 3 + d]; g[b * 6 + 3 + d] = fb[nb * 3 + b * 3 + d]; }
    }
  return run_adjoint(h, G, grads, stats);
}

int dfx_objective_kinetic(dfx_handle* h, const int32_t* target_blocks, int32_t n_target, double* objective) {
  const Plan& pl = h->pl;
  const int nb = pl.n_blocks, B = pl.batch, Tn = h->n_tp;
  if (h->fields.empty()) { h->err = "objective: run forward first"; return 1; }
  for (int m = 0; m < B; ++m) {
    double acc = 0.0;
    for (int k = 0; k < Tn; ++k)
      for (int i = 0; i < n_target; ++i)
        for (int d = 0; d < 3; ++d) {
          int dof = target_blocks[i] * 3 + d;
          double v = h->fields[((size_t)m * Tn + k) * nb * 6 + nb * 3 + dof];
          acc += 0.5 * v * v / h->pp.inv_m[(size_t)m * nb * 3 + dof];
        }
    objective[m] = acc;
  }
  return 0;
}

int dfx_adjoint_kinetic(dfx_handle* h, const int32_t* target_blocks, int32_t n_target, dfx_grads* grads, dfx_stats* stats) {
  const Plan& pl = h->pl;
  const int nb = pl.n_blocks, B = pl.batch, Tn = h->n_tp;
  if (h->fields.empty()) { h->err = "adjoint_kinetic: run forward first"; return 1; }
  std::vector<double> G((size_t)B * Tn * nb * 6, 0.0);
  for (int m = 0; m < B; ++m)
    for (int k = 0; k < Tn; ++k)
      for (int i = 0; i < n_target; ++i)
        for (int d = 0; d < 3; ++d) {
          int b = target_blocks[i], dof = b * 3 + d;
          double v = h->fields[((size_t)m * Tn + k) * nb * 6 + nb * 3 + dof];
          G[((size_t)m * Tn + k) * nb * 6 + b * 6 + 3 + d] = v / h->pp.inv_m[(size_t)m * nb * 3 + dof];
        }
  int rc = run_adjoint(h, G, grads, stats);
  if (rc) return rc;
  // explicit dependence of the objective on the inertia: d/dm sum m v^2/2 = v^2/2
  if (grads && grads->inertia)
    for (int m = 0; m < B; ++m)
      for (int k = 0; k < Tn; ++k)
        for (int i = 0; i < n_target; ++i)
          for (int d = 0; d < 3; ++d) {
            int dof = target_blocks[i] * 3 + d;
            double v = h->fields[((size_t)m * Tn + k) * nb * 6 + nb * 3 + dof];
            grads->inertia[(size_t)m * nb * 3 + dof] += 0.5 * v * v;
          }
  return 0;
}

int dfx_kinetic_value_and_grad(dfx_handle* h, const int32_t* target_blocks, int32_t n_target, double* objective,
                               const dfx_grads* want, dfx_grads* views, dfx_stats* stats) {
  // same contract as the HIP engine: results in handle-owned memory, valid until the next call on the handle
  const Plan& pl = h->pl;
  const size_t B = pl.batch, nb = pl.n_blocks, nbd = pl.n_bonds;
  const size_t sizes[10] = {B * nb * pl.n_npb * 2, B * nbd * 2, B * nbd * 3, B * nb * 3, B * nb * 3, B * nbd * 2, B * 3,
                            B * (size_t)std::max(1, pl.n_fns) * DFX_FN_PARAMS, B * nb * 6, B * nb * 2};
  dfx_grads g;
  double** gp = reinterpret_cast<double**>(&g);
  double* const* wp = reinterpret_cast<double* const*>(want);
  for (int i = 0; i < 10; ++i) {
    gp[i] = nullptr;
    if (want && wp[i]) { h->view_store[i].assign(sizes[i], 0.0); gp[i] = h->view_store[i].data(); }
  }
  if (objective) if (int rc = dfx_objective_kinetic(h, target_blocks, n_target, objective)) return rc;
  if (int rc = dfx_adjoint_kinetic(h, target_blocks, n_target, &g, stats)) return rc;
  if (views) *views = g;
  return 0;
}

int dfx_response_data(dfx_handle* h, double* e_stretch, double* e_shear, double* e_bend, double* e_kin) {
  const Plan& pl = h->pl;
  const int nb = pl.n_blocks, B = pl.batch, Tn = h->n_tp, nbd = pl.n_bonds;
  if (h->fields.empty()) { h->err = "response_data: run forward first"; return 1; }
  for (int m = 0; m < B; ++m) {
    Tables tb = member_tables(h, m);
    for (int k = 0; k < Tn; ++k) {
      const double* f = h->fields.data() + ((size_t)m * Tn + k) * nb * 6;
      if (e_kin)
        for (int b = 0; b < nb; ++b) {
          double acc = 0.0;
          for (int d = 0; d < 3; ++d) acc += 0.5 * f[nb * 3 + b * 3 + d] * f[nb * 3 + b * 3 + d] / tb.inv_m[b * 3 + d];
          e_kin[((size_t)m * Tn + k) * nb + b] = acc;
        }
      for (int s = 0; s < pl.n_slots; ++s) {
        LigRef lr;
        for (int which = 0; slot_ligament(tb, s, which, lr); ++which) {
          const int info = lr.info;
          if (info & 1) continue;                                   // every ligament once: on its end-0 side
          const int ps = info >> 1;
          BlockRec<double> o, p;
          const double* uo = f + (size_t)(s >> 2) * 3; const double* up = f + (size_t)(ps >> 2) * 3;
          o.x = uo[0]; o.y = uo[1]; o.th = uo[2]; o.ch = cos(0.5 * uo[2]); o.sh = sin(0.5 * uo[2]);
          p.x = up[0]; p.y = up[1]; p.th = up[2]; p.ch = cos(0.5 * up[2]); p.sh = sin(0.5 * up[2]);
          const double* sp = tb.slot_p + (size_t)s * kSlotParams; const double* pp = tb.slot_p + (size_t)ps * kSlotParams;
          const double* bp = lr.bp;
          const double l0 = sqrt(bp[0] * bp[0] + bp[1] * bp[1]);
          BondGrad<double> g;
          bond_grad<kNonlinear, double>(o, p, sp[0], sp[1], pp[0], pp[1], bp[0], bp[1], l0, 1.0 / l0, bp[2], bp[3], bp[4], -1.0, g);
          const size_t oi = ((size_t)m * Tn + k) * nbd + (lr.ovf >= 0 ? pl.ovf_bond[lr.ovf] : pl.slot_bond[s]);
          if (e_stretch) e_stretch[oi] = bp[2] * g.ks;
          if (e_shear) e_shear[oi] = bp[3] * g.ksh;
          if (e_bend) e_bend[oi] = bp[4] * g.kr;
        }
      }
    }
  }
  return 0;
}

int dfx_rhs(dfx_handle* h, const double* y, double t, double* dy) {
  if (!h->have_params) { h->err = "rhs: set_params first"; return 1; }
  const Plan& pl = h->pl;
  const int nb = pl.n_blocks;
  std::vector<double> S((size_t)nb * kRec), A((size_t)pl.tab.s * nb * 3);
  for (int m = 0; m < pl.batch; ++m) {
    Tables tb = member_tables(h, m);
    for (int b = 0; b < nb; ++b)
      for (int d = 0; d < 3; ++d) init_dof(tb, y + (size_t)m * nb * 6, t, S.data(), b, d);
    FwdStage st;
    st.S_in = S.data(); st.S_out = nullptr; st.Y = S.data(); st.A = A.data(); st.i = 0; st.h = 0.0; st.t_i = t; st.t_next = t;
    fwd_stage(tb, pl.tab, st);
    double* o = dy + (size_t)m * nb * 6;
    for (int b = 0; b < nb; ++b)
      for (int d = 0; d < 3; ++d) {
        int sidx = pl.block_special[b];
        bool con = sidx >= 0 && ((pl.special[sidx].con_mask >> d) & 1);
        o[b * 3 + d] = con ? 0.0 : S[(size_t)b * kRec + 5 + d];
        o[nb * 3 + b * 3 + d] = A[b * 3 + d];
      }
  }
  return 0;
}

int dfx_rhs_vjp(dfx_handle* h, const double* y, double t, const double* lam, double* y_bar, dfx_grads* grads) {
  if (!h->have_params) { h->err = "rhs_vjp: set_params first"; return 1; }
  const Plan& pl = h->pl;
  const int nb = pl.n_blocks, B = pl.batch;
  const int nsp = pl.n_special > 0 ? pl.n_special : 1;
  std::vector<double> slot_g((size_t)B * pl.n_slots * kSlotGrads, 0.0), blk_g((size_t)B * nb * 6, 0.0),
      fn_g((size_t)B * nsp * DFX_MAX_FNS * DFX_FN_PARAMS, 0.0), cen_g((size_t)B * nb * 2, 0.0),
      ovf_g((size_t)B * pl.n_ovf * kOvfGrads + 1, 0.0);
  std::vector<double> S((size_t)nb * kRec), A((size_t)pl.tab.s * nb * 3), W((size_t)nb * 3);
  for (int m = 0; m < B; ++m) {
    Tables tb = member_tables(h, m);
    GradAcc acc{slot_g.data() + (size_t)m * pl.n_slots * kSlotGrads, blk_g.data() + (size_t)m * nb * 6,
                fn_g.data() + (size_t)m * nsp * DFX_MAX_FNS * DFX_FN_PARAMS, cen_g.data() + (size_t)m * nb * 2,
                ovf_g.data() + (size_t)m * pl.n_ovf * kOvfGrads};
    for (int b = 0; b < nb; ++b)
      for (int d = 0; d < 3; ++d) init_dof(tb, y + (size_t)m * nb * 6, t, S.data(), b, d);
    FwdStage st;
    st.S_in = S.data(); st.S_out = nullptr; st.Y = S.data(); st.A = A.data(); st.i = 0; st.h = 0.0; st.t_i = t; st.t_next = t;
    fwd_stage(tb, pl.tab, st);
    const double* lm = lam + (size_t)m * nb * 6;
    for (int b = 0; b < nb; ++b)
      for (int d = 0; d < 3; ++d) {
        int sidx = pl.block_special[b];
        bool con = sidx >= 0 && ((pl.special[sidx].con_mask >> d) & 1);
        W[b * 3 + d] = con ? 0.0 : lm[nb * 3 + b * 3 + d] * tb.inv_m[b * 3 + d];
      }
    double* yb = y_bar + (size_t)m * nb * 6;
    for (int b = 0; b < nb; ++b) {
      double hs[3] = {0, 0, 0};
      for (int k = 0; k < kSlots; ++k) {
        double hx = 0.0, hy = 0.0, hth = 0.0;
#define CASE(M) case M: if (tb.contact == 2) adj_slot<M, 2>(tb, S.data(), W.data(), b * kSlots + k, acc, hx, hy, hth); else if (tb.contact) adj_slot<M, 1>(tb, S.data(), W.data(), b * kSlots + k, acc, hx, hy, hth); else adj_slot<M, 0>(tb, S.data(), W.data(), b * kSlots + k, acc, hx, hy, hth); break;
        switch (tb.model) { CASE(kNonlinear) CASE(kLinearized) CASE(kSimpleSpring) CASE(kStretchTorsion) }
#undef CASE
        hs[0] += hx; hs[1] += hy; hs[2] += hth;
      }
      for (int d = 0; d < 3; ++d) {
        bool con; double ybq, ybv;
        adj_dof_local(tb, acc, b, d, t, hs[d], W[b * 3 + d], lm[b * 3 + d], S[(size_t)b * kRec + 5 + d], A[b * 3 + d], con, ybq, ybv);
        yb[b * 3 + d] = ybq;
        yb[nb * 3 + b * 3 + d] = ybv;
      }
    }
  }
  if (grads) {
    dfx_grads g = *grads;
    g.state0 = nullptr;
    unpack_grads(pl, slot_g, blk_g, fn_g, h->pp.inv_m, &g, &ovf_g);
    if (g.block_centroids) memcpy(g.block_centroids, cen_g.data(), sizeof(double) * cen_g.size());
  }
  return 0;
}

int dfx_energy(dfx_handle* h, const double* u, double* energy) {
  if (!h->have_params) { h->err = "energy: set_params first"; return 1; }
  const Plan& pl = h->pl;
  const int nb = pl.n_blocks;
  std::vector<double> S((size_t)nb * kRec, 0.0);
  for (int m = 0; m < pl.batch; ++m) {
    Tables tb = member_tables(h, m);
    for (int b = 0; b < nb; ++b) {
      double* r = S.data() + (size_t)b * kRec;
      for (int d = 0; d < 3; ++d) r[d] = u[(size_t)m * nb * 3 + b * 3 + d];
      r[3] = cos(0.5 * r[2]); r[4] = sin(0.5 * r[2]);
    }
    FwdStage st;
    st.S_in = S.data(); st.S_out = nullptr; st.Y = S.data(); st.A = nullptr; st.i = 0; st.h = 0; st.t_i = 0; st.t_next = 0;
    fwd_stage(tb, pl.tab, st, energy + m);
  }
  return 0;
}

// failure isolation (include/dfx.h): the port judges a member by its output rows
int dfx_member_status(dfx_handle* h, int32_t* status) {
  const Plan& pl = h->pl;
  const size_t per = h->fields.size() / std::max(1, pl.batch);
  for (int m = 0; m < pl.batch; ++m) {
    status[m] = 0;
    for (size_t i = 0; i < per && !h->fields.empty(); ++i) if (!std::isfinite(h->fields[(size_t)m * per + i])) { status[m] = 1; break; }
  }
  return 0;
}
int dfx_set_failure_policy(dfx_handle*, int32_t) { return 0; }      // (the port never fails a fixed-grid call for a non-finite member)
int dfx_test_set_spin_limit(dfx_handle*, int32_t) { return 0; }
// design -> geometry and its cotangent (host code shared with libdfx: dfx_design.h)
int dfx_design_forward(const dfx_design_map* map, const double* design, int32_t batch, double density, double* block_centroids,
                       double* centroid_node_vectors, double* inertia, double* void_angle0) {
  return dfx_design::forward(map, design, batch, density, block_centroids, centroid_node_vectors, inertia, void_angle0);
}
int dfx_design_vjp(const dfx_design_map* map, const double* design, int32_t batch, double density, const double* centroid_node_vectors_bar,
                   const double* block_centroids_bar, const double* inertia_bar, const double* void_angle0_bar, double* design_bar) {
  return dfx_design::vjp(map, design, batch, density, centroid_node_vectors_bar, block_centroids_bar, inertia_bar, void_angle0_bar, design_bar);
}
int dfx_device_count(void) { return 0; }
const char* dfx_version(void) { return "dfx-cpu-port 0.1.0"; }
int dfx_abi_layout(int32_t* out, int32_t n) { return dfxabi_fill(out, n); }

}  // extern "C"
