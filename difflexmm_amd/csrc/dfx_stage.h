// dfx_stage.h -- what one lane does in one Runge-Kutta stage, forward and reverse.
//
// Written once as host/device functions over plain pointers: dfx_engine.hip wraps them in
// gfx950 kernels (one lane per (block, node slot), quad shuffles for the 4-slot sums), the CPU
// port (oracle/cpu/dfx_cpu.cpp) wraps them in loops.
//
// Forward stage i of a step (reference: rhs, difflexmm/dynamics.py:33-55, one evaluation inside
// runge_kutta_step of jax.experimental.ode):
//   slot part : dE/d(own block DOFs) of the ligament (+ contact) attached to this node slot
//   DOF part  : A_i = (-dE/du + F_load(t_i) - c v) / m   (0 for constrained DOFs), then the NEXT stage
//               record  Q_{i+1} = q_n + h c_{i+1} v_n + h^2 sum_l aa_{i+1,l} A_l,
//                       V_{i+1} = v_n + h sum_l a_{i+1,l} A_l   (constrained DOFs: c(t_{i+1}), c'(t_{i+1}))
//               so the next launch only gathers finished records (kinematics.py:68-77 folded in).
// Reverse stage i (discrete adjoint of the same step; replaces jax's continuous _odeint_rev):
//   slot part : Hessian-vector product (H w)_own and the mixed parameter derivatives by pushing Dual
//               numbers seeded with w = kbar_v / m through bond_grad / contact_grad
//   DOF part  : Ybar_i = (-H w, kbar_q - c w); parameter accumulators; then kbar_{i-1} for the next
//               launch, or at i = 0 the step's lambda_n.
#pragma once
#include "dfx_physics.h"
#include "dfx_plan.h"

namespace dfx {

// per-member device/host image of the static tables + packed parameters
struct Tables {
  int n_blocks, n_fns, model, contact, n_npb;
  const int32_t* slot_info;
  const int32_t* block_special;
  const dfx_special* special;
  const double* slot_p;    // n_slots * kSlotParams
  const double* inv_m;     // n_blocks * 3
  const double* damping;   // n_blocks * 3
  const double* contact_p; // 3
  const double* centroid;  // n_blocks * 2 (distance-based contact)
  const TimeFn* fns;       // DFX_MAX_FNS
  // extra ligaments of nodes that carry more than one (Plan::ovf_*), or null
  const int32_t* ovf_ptr = nullptr;
  const int32_t* ovf_info = nullptr;
  const double* ovf_p = nullptr;     // n_ovf * kOvfParams
};

// Ligament number `which` of a slot: 0 = the slot's first one (slot_info / slot_p), 1.. = its extra ones.  Returns false past the end.
struct LigRef { int info; const double* bp; /* l0(2) k(3) phi(2) */ int ovf; /* entry in the overflow list or -1 */ };
DFX_HD bool slot_ligament(const Tables& tb, int slot, int which, LigRef& r) {
  if (which == 0) {
    r.info = tb.slot_info[slot]; r.bp = tb.slot_p + (size_t)slot * 9 + 2; r.ovf = -1;
    return r.info >= 0;
  }
  if (!tb.ovf_ptr) return false;
  const int e = tb.ovf_ptr[slot] + which - 1;
  if (e >= tb.ovf_ptr[slot + 1]) return false;
  r.info = tb.ovf_info[e]; r.bp = tb.ovf_p + (size_t)e * 8; r.ovf = e;
  return true;
}

DFX_HD BlockRec<double> load_rec(const double* S, int b) {
  const double* r = S + (size_t)b * kRec;
  BlockRec<double> o;
  o.x = r[0]; o.y = r[1]; o.th = r[2]; o.ch = r[3]; o.sh = r[4];
  return o;
}

// distance-based contact (CONTACT == 2): node vectors of the bonded node, its next and its previous node on the block
DFX_HD void node_triple(const Tables& tb, int slot, double (&r)[3][2]) {
  const int b = slot >> 2, k = slot & 3, n = tb.n_npb;
  const int ks[3] = {k, (k + 1) % n, (k + n - 1) % n};
  for (int i = 0; i < 3; ++i) {
    const double* sp = tb.slot_p + (size_t)(b * kSlots + ks[i]) * kSlotParams;
    r[i][0] = sp[0]; r[i][1] = sp[1];
  }
}

// ---- forward -------------------------------------------------------------------------------
template <int MODEL, int CONTACT>
DFX_HD void fwd_slot(const Tables& tb, const double* S_in, int slot, double& fx, double& fy, double& fth, double* energy) {
  fx = 0.0; fy = 0.0; fth = 0.0;
  if (energy) *energy = 0.0;
  LigRef lr;
  for (int which = 0; slot_ligament(tb, slot, which, lr); ++which) {
  const int info = lr.info;
  int ps = info >> 1;
  double sgn = (info & 1) ? 1.0 : -1.0;
  const double* sp = tb.slot_p + (size_t)slot * kSlotParams;
  const double* pp = tb.slot_p + (size_t)ps * kSlotParams;
  const double* bp = lr.bp;        // l0(2) k(3) phi(2) of THIS ligament
  BlockRec<double> o = load_rec(S_in, slot >> 2), p = load_rec(S_in, ps >> 2);
  BondGrad<double> g;
  const double l0 = sqrt(bp[0] * bp[0] + bp[1] * bp[1]);
  bond_grad<MODEL, double>(o, p, sp[0], sp[1], pp[0], pp[1], bp[0], bp[1], l0, 1.0 / l0, bp[2], bp[3], bp[4], sgn, g);
  fx += g.fx; fy += g.fy; fth += g.fth;
  double e = g.e;
  if (CONTACT == 2) {
    double ro[3][2], rp[3][2];
    node_triple(tb, slot, ro);
    node_triple(tb, ps, rp);
    DistContactGrad<double> c;
    distance_contact_grad<double, double>(o, p, tb.centroid[(slot >> 2) * 2], tb.centroid[(slot >> 2) * 2 + 1], tb.centroid[(ps >> 2) * 2],
                                          tb.centroid[(ps >> 2) * 2 + 1], ro, rp, info & 1, tb.contact_p[0], tb.contact_p[1], tb.contact_p[2], c);
    fx += c.fx; fy += c.fy; fth += c.fth;
    e += c.e;
  } else if (CONTACT) {
    ContactGrad<double> c;
    double kap = sgn * (o.th - p.th);
    contact_grad<double>(kap, bp[5], bp[6], tb.contact_p[0], tb.contact_p[1], tb.contact_p[2], c);
    fth += sgn * c.dkap;
    e += c.e;
  }
  if (energy && !(info & 1)) *energy += e;  // every ligament is counted once, on its end-0 side
  }
}

struct FwdStage {
  const double* S_in;   // stage records of this stage
  const double* Y;      // step base records (q_n, v_n)
  double* A;            // accelerations, stage-major: A[l * n_blocks*3 + dof]
  double* S_out;        // next stage records (may be null)
  int i;                // stage index 0..s-1
  double t_i, t_next, h;
};

// DOF d of block b; dE = sum over the block's slots of dE/du_d
DFX_HD void fwd_dof(const Tables& tb, const Tableau& T, const FwdStage& st, int b, int d, double dE) {
  const int nd = tb.n_blocks * 3;
  const int dof = b * 3 + d;
  int sidx = tb.block_special[b];
  bool constrained = false;
  double fload = 0.0, cnext = 0.0, cdnext = 0.0;
  if (sidx >= 0) {
    const dfx_special& sp = tb.special[sidx];
    constrained = (sp.con_mask >> d) & 1;
    double gp[kMaxFnParams];
    for (int f = 0; f < tb.n_fns; ++f) {
      double g, gt;
      if (constrained) {
        if (sp.con_coef[d][f] != 0.0) {
          eval_time_fn(tb.fns[f], st.t_next, g, gt, gp);
          cnext += sp.con_coef[d][f] * g;
          cdnext += sp.con_coef[d][f] * gt;
        }
      } else if (sp.load_coef[d][f] != 0.0) {
        eval_time_fn(tb.fns[f], st.t_i, g, gt, gp);
        fload += sp.load_coef[d][f] * g;
      }
    }
  }
  const double* rin = st.S_in + (size_t)b * kRec;
  double v_i = rin[5 + d];
  double a = constrained ? 0.0 : (fload - dE - tb.damping[dof] * v_i) * tb.inv_m[dof];
  st.A[(size_t)st.i * nd + dof] = a;
  if (!st.S_out) return;
  const double* yb = st.Y + (size_t)b * kRec;
  double qn = yb[d], vn = yb[5 + d];
  const int r = st.i + 1;  // row of the tableau for the next stage (row s = solution weights)
  double sv = T.a[r][st.i] * a, sq = T.aa[r][st.i] * a;
  for (int l = 0; l < st.i; ++l) {
    double al = st.A[(size_t)l * nd + dof];
    sv += T.a[r][l] * al;
    sq += T.aa[r][l] * al;
  }
  double qnext = qn + st.h * (T.c[r] * vn + st.h * sq);
  double vnext = vn + st.h * sv;
  if (constrained) { qnext = cnext; vnext = cdnext; }
  double* ro = st.S_out + (size_t)b * kRec;
  ro[d] = qnext;
  ro[5 + d] = vnext;
  if (d == 2) {
    double s, c;
    fast_sincos(0.5 * qnext, &s, &c);
    ro[3] = c;
    ro[4] = s;
  }
}

// Build the record of the initial state (row 0 of `fields`): constrained DOFs follow c(t0), c'(t0).
DFX_HD void init_dof(const Tables& tb, const double* state0 /* (2, n_blocks, 3) */, double t0, double* S, int b, int d) {
  const int nd = tb.n_blocks * 3;
  double q = state0[b * 3 + d], v = state0[nd + b * 3 + d];
  int sidx = tb.block_special[b];
  if (sidx >= 0 && ((tb.special[sidx].con_mask >> d) & 1)) {
    q = 0.0; v = 0.0;
    double gp[kMaxFnParams];
    for (int f = 0; f < tb.n_fns; ++f) {
      double g, gt;
      eval_time_fn(tb.fns[f], t0, g, gt, gp);
      q += tb.special[sidx].con_coef[d][f] * g;
      v += tb.special[sidx].con_coef[d][f] * gt;
    }
  }
  double* r = S + (size_t)b * kRec;
  r[d] = q;
  r[5 + d] = v;
  if (d == 2) {
    double s, c;
    fast_sincos(0.5 * q, &s, &c);
    r[3] = c;
    r[4] = s;
  }
}

// ---- reverse -------------------------------------------------------------------------------
struct GradAcc {
  double* slot_g;  // n_slots * kSlotGrads (or null)
  double* blk_g;   // n_blocks * 6        (or null)
  double* fn_g;    // n_special * DFX_MAX_FNS * DFX_FN_PARAMS (or null)
  double* cen_g;   // n_blocks * 2: d/d(block_centroids) (distance-based contact; or null)
  double* ovf_g = nullptr;   // n_ovf * kOvfGrads: the bond part of the extra ligaments (same layout as a slot's entries 2..11), or null
};

// Own-side Hessian-vector product of slot `slot` for direction W (n_blocks*3) at records S.
template <int MODEL, int CONTACT>
DFX_HD void adj_slot(const Tables& tb, const double* S, const double* W, int slot, const GradAcc& acc,
                     double& hx, double& hy, double& hth) {
  hx = 0.0; hy = 0.0; hth = 0.0;
  LigRef lr;
  for (int which = 0; slot_ligament(tb, slot, which, lr); ++which) {
  const int info = lr.info;
  int ps = info >> 1;
  double sgn = (info & 1) ? 1.0 : -1.0;
  int bo = slot >> 2, bp = ps >> 2;
  const double* sp = tb.slot_p + (size_t)slot * kSlotParams;
  const double* pp = tb.slot_p + (size_t)ps * kSlotParams;
  const double* lp = lr.bp;        // l0(2) k(3) phi(2) of THIS ligament
  BlockRec<double> ro = load_rec(S, bo), rp = load_rec(S, bp);
  BlockRec<Dual> o = seed_rec(ro, W[bo * 3], W[bo * 3 + 1], W[bo * 3 + 2]);
  BlockRec<Dual> p = seed_rec(rp, W[bp * 3], W[bp * 3 + 1], W[bp * 3 + 2]);
  BondGrad<Dual> g;
  const double l0 = sqrt(lp[0] * lp[0] + lp[1] * lp[1]);
  bond_grad<MODEL, Dual>(o, p, sp[0], sp[1], pp[0], pp[1], lp[0], lp[1], l0, 1.0 / l0, lp[2], lp[3], lp[4], sgn, g);
  hx += g.fx.e; hy += g.fy.e; hth += g.fth.e;
  ContactGrad<Dual> c;
  DistContactGrad<Dual> dc;
  if (CONTACT == 2) {
    double ro[3][2], rpp[3][2];
    node_triple(tb, slot, ro);
    node_triple(tb, ps, rpp);
    distance_contact_grad<Dual, double>(o, p, tb.centroid[bo * 2], tb.centroid[bo * 2 + 1], tb.centroid[bp * 2], tb.centroid[bp * 2 + 1], ro, rpp,
                                        info & 1, tb.contact_p[0], tb.contact_p[1], tb.contact_p[2], dc);
    hx += dc.fx.e; hy += dc.fy.e; hth += dc.fth.e;
    if (acc.slot_g) {
      const int k = slot & 3, n = tb.n_npb, ks[3] = {k, (k + 1) % n, (k + n - 1) % n};
      for (int i = 0; i < 3; ++i) {       // the three nodes sit on the own block: same thread in the block loop of the CPU port
        double* q = acc.slot_g + (size_t)(bo * kSlots + ks[i]) * kSlotGrads;
        q[0] -= dc.r[i][0].e; q[1] -= dc.r[i][1].e;
      }
      if (!(info & 1)) {
        double* q = acc.slot_g + (size_t)slot * kSlotGrads;
        q[9] -= dc.am.e; q[10] -= dc.ac.e; q[11] -= dc.kc.e;
      }
    }
    if (acc.cen_g) { acc.cen_g[bo * 2] -= dc.cx.e; acc.cen_g[bo * 2 + 1] -= dc.cy.e; }
  } else if (CONTACT) {
    Dual kap = sgn * (o.th - p.th);
    contact_grad<Dual>(kap, lp[5], lp[6], tb.contact_p[0], tb.contact_p[1], tb.contact_p[2], c);
    hth += sgn * c.dkap.e;
  }
  if (acc.slot_g) {
    // L += w . F = -w . grad E   =>   dL/dp = -eps(dE/dp)
    double* q = acc.slot_g + (size_t)slot * kSlotGrads;
    q[0] -= g.rx.e;
    q[1] -= g.ry.e;
    if (lr.ovf >= 0) q = acc.ovf_g + (size_t)lr.ovf * kOvfGrads;      // the bond part of an extra ligament has its own entry
    if (!(info & 1)) {
      q[2] -= g.lx.e; q[3] -= g.ly.e;
      q[4] -= g.ks.e; q[5] -= g.ksh.e; q[6] -= g.kr.e;
      if (CONTACT == 1) {
        q[7] -= c.p1.e; q[8] -= c.p2.e;
        q[9] -= c.am.e; q[10] -= c.ac.e; q[11] -= c.kc.e;
      }
    }
  }
  }
}

struct AdjStage {
  const double* S;      // records of stage i (recomputed)
  const double* A;      // accelerations of all stages of this step (stage-major)
  const double* W;      // w_i = kbar_v,i / m   (n_blocks*3), 0 on constrained DOFs
  const double* KQ;     // kbar_q,i            (n_blocks*3)
  double* YB;           // Ybar, stage-major: YB[j * n_blocks*6 + b*6 + {0..2: q, 3..5: v}]
  double* LAM;          // lambda_{n+1} on entry; lambda_n written at i == 0   (n_blocks*6)
  double* W_out;        // w for the next launch (stage i-1, or stage s-1 of the previous step)
  double* KQ_out;
  const double* G;      // output cotangent to add to lambda_n at i == 0 (n_blocks*6) or null
  int i;
  int local_only;       // 1: only Ybar_i + parameter accumulators (dfx_rhs_vjp test hook)
  double t_i, h, h_prev;  // h_prev: step size of the step that the NEXT reverse launch belongs to
  // dense-output adjoint (the adaptive solve's outputs are interpolated inside a step: dfx_forward_adaptive_keep): what the outputs
  // inside this step (and, through the FSAL slope, inside the previous one) add to the Kbar this launch hands on, already scaled by the
  // step size -- (n_blocks*6: q(3) v(3) per block) or null
  const double* src = nullptr;
};

// Local part of the reverse DOF work: Ybar of this DOF and the parameter accumulators.
//   hw = (H w)_dof, w = kbar_v/m, kq = kbar_q, v_i / a_i = stage velocity / acceleration of the DOF.
DFX_HD void adj_dof_local(const Tables& tb, const GradAcc& acc, int b, int d, double t_i, double hw, double w,
                          double kq, double v_i, double a_i, bool& constrained, double& ybq, double& ybv) {
  const int dof = b * 3 + d;
  int sidx = tb.block_special[b];
  constrained = false;
  if (sidx >= 0) {
    const dfx_special& sp = tb.special[sidx];
    constrained = (sp.con_mask >> d) & 1;
    if (acc.fn_g) {
      double gp[kMaxFnParams];
      for (int f = 0; f < tb.n_fns; ++f) {
        double coef = constrained ? -hw * sp.con_coef[d][f] : w * sp.load_coef[d][f];
        if (coef != 0.0) {
          double g, gt;
          eval_time_fn(tb.fns[f], t_i, g, gt, gp);
          double* q = acc.fn_g + ((size_t)sidx * DFX_MAX_FNS + f) * DFX_FN_PARAMS;
          for (int k = 0; k < DFX_FN_PARAMS; ++k) q[k] += coef * gp[k];
        }
      }
    }
  }
  ybq = 0.0; ybv = 0.0;
  if (!constrained) {
    ybq = -hw;
    ybv = kq - tb.damping[dof] * w;
    if (acc.blk_g) {
      double* q = acc.blk_g + (size_t)b * 6;
      q[d] -= w * a_i;
      q[3 + d] -= w * v_i;
    }
  }
}

DFX_HD void adj_dof(const Tables& tb, const Tableau& T, const AdjStage& st, const GradAcc& acc, int b, int d, double hw) {
  const int nd = tb.n_blocks * 3;
  const int nd6 = tb.n_blocks * 6;
  const int dof = b * 3 + d;
  bool constrained;
  double ybq, ybv;
  adj_dof_local(tb, acc, b, d, st.t_i, hw, st.W[dof], st.KQ[dof], st.S[(size_t)b * kRec + 5 + d],
                st.A[(size_t)st.i * nd + dof], constrained, ybq, ybv);
  st.YB[(size_t)st.i * nd6 + b * 6 + d] = ybq;
  st.YB[(size_t)st.i * nd6 + b * 6 + 3 + d] = ybv;
  if (st.local_only) return;
  double lq = st.LAM[b * 6 + d], lv = st.LAM[b * 6 + 3 + d];
  double kq, kv;
  if (st.i > 0) {
    // kbar_{i-1} = h b_{i-1} lambda_{n+1} + h sum_{j >= i} a_{j,i-1} Ybar_j
    const int c = st.i - 1;
    double sq = T.a[T.s][c] * lq, sv = T.a[T.s][c] * lv;
    sq += T.a[st.i][c] * ybq;
    sv += T.a[st.i][c] * ybv;
    for (int j = st.i + 1; j < T.s; ++j) {
      sq += T.a[j][c] * st.YB[(size_t)j * nd6 + b * 6 + d];
      sv += T.a[j][c] * st.YB[(size_t)j * nd6 + b * 6 + 3 + d];
    }
    kq = st.h * sq;
    kv = st.h * sv;
  } else {
    // lambda_n = lambda_{n+1} + sum_j Ybar_j (+ output cotangent); then kbar_{s-1} of step n-1
    lq += ybq; lv += ybv;
    for (int j = 1; j < T.s; ++j) {
      lq += st.YB[(size_t)j * nd6 + b * 6 + d];
      lv += st.YB[(size_t)j * nd6 + b * 6 + 3 + d];
    }
    if (st.G && !constrained) { lq += st.G[b * 6 + d]; lv += st.G[b * 6 + 3 + d]; }
    if (constrained) { lq = 0.0; lv = 0.0; }
    st.LAM[b * 6 + d] = lq;
    st.LAM[b * 6 + 3 + d] = lv;
    const double bw = T.a[T.s][T.s - 1];
    kq = st.h_prev * bw * lq;
    kv = st.h_prev * bw * lv;
  }
  if (st.src && !constrained) { kq += st.src[b * 6 + d]; kv += st.src[b * 6 + 3 + d]; }
  st.KQ_out[dof] = kq;
  st.W_out[dof] = constrained ? 0.0 : kv * tb.inv_m[dof];
}

// Start of the reverse sweep: lambda_N = G_last; kbar_{s-1} of the last step.
DFX_HD void adj_begin_dof(const Tables& tb, const Tableau& T, const double* G, double h_last, double* LAM,
                          double* W_out, double* KQ_out, int b, int d) {
  int sidx = tb.block_special[b];
  bool constrained = sidx >= 0 && ((tb.special[sidx].con_mask >> d) & 1);
  double lq = constrained ? 0.0 : G[b * 6 + d], lv = constrained ? 0.0 : G[b * 6 + 3 + d];
  LAM[b * 6 + d] = lq;
  LAM[b * 6 + 3 + d] = lv;
  const double bw = T.a[T.s][T.s - 1];
  KQ_out[b * 3 + d] = h_last * bw * lq;
  W_out[b * 3 + d] = constrained ? 0.0 : h_last * bw * lv * tb.inv_m[b * 3 + d];
}

}  // namespace dfx
