// dfx_persist_dense.h -- the reference's own integrator inside the persistent stage loop (round 6): jax.experimental.ode.odeint as called at
// difflexmm/dynamics.py:166 -- adaptive Dormand-Prince 5(4), one controller per ensemble member -- without a kernel boundary between its
// evaluations, and the reverse sweep of the steps it accepted (dense-output discrete adjoint, dfx_dense.h) in the same form.
//
// Forward (k_adaptive_fwd_loop).  A wave owns its 16 blocks for a whole run of attempts, as in k_fwd_persist (dfx_persist.h): step state, the
// seven stage accelerations (A_0 is the FSAL slope) and the ligament's parameters in registers, the 32-byte stage records of its blocks
// through the hand-off ring (6 per attempt: S_1 .. S_5 and the candidate y1; the data is the flag).  What is new is the step controller:
//   * error norm: every wave reduces the squared error ratios of its DOFs in the order of k_fwd_stage's error mode (shfl_down tree) and
//     publishes ONE double per attempt into err[attempt % 3][member][wave] (write-through; poison until written); every wave of the
//     member then reads ALL the member's partials (an all-gather: one more hand-off per attempt, not per stage) and adds them in the
//     order of k_control (thread t of 128 takes partials t, t + 128, ..; tree over the threads) -- so every wave holds the same bits
//     of `ratio`, takes the same accept / reject decision and computes the same next step without a broadcast, and in a build without
//     floating-point contraction the sequence of steps equals the stage-launch controller's bit for bit;
//   * a slot of the partials is re-poisoned by its owner right after it has completed the gather of the NEXT attempt (every wave has then
//     posted that attempt's partial, i.e. has finished reading this one's), two attempts before the slot is due again: the polls of the
//     six records in between wait for that store (s_waitcnt vmcnt(0)) long before the owner posts into the slot;
//   * dense output (jax's quartic, dopri_dense) for the outputs an accepted step crosses, straight into the resident history; commit =
//     a register move (y1 -> y_n, A_6 -> A_0); wave 0 of a member records what the controller of the stage launches records (clock,
//     accepted step times, steps per output interval, and -- when the steps are kept for the reverse sweep -- AdaptRec);
//   * keep: the attempt's records go to the trajectory checkpoint of step `accepted` as well (plain stores beside the ring, the layout the
//     stage launches write), a rejected attempt's records are overwritten by the next.
// The launch ends when every member has produced its last output, after pa.n_steps attempts, or when the room for kept steps is used up
// (the host grows it and launches again: the state a launch leaves behind -- clock, y_n, A_0 -- is what the next one, or the stage
// launches, start from).
#pragma once
#include "dfx_persist.h"
#include "dfx_dense.h"

// ---- wave-wide helpers of the controller.  The sums keep the ORDER of the shfl_down tree (32, 16, .. 1) the stage launches use, so the bits
// agree, but not its cost: a 64-bit __shfl_down is two ds_bpermute round trips per level (~0.3 us for the tree, twice per attempt, on every
// wave's critical path); gfx950's permlane swaps bring lanes i+32 / i+16 to lane i, DPP row shifts the rest (tools/mock/permlane_probe.hip:
// identical bits)
template <int CTRL> __device__ __forceinline__ double dpp_row_mov(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lanes_plus_32(double v) {
  const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double((int)b[1], (int)a[1]);
}
__device__ __forceinline__ double lanes_plus_16(double v) {
  const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  return __hiloint2double((int)b[1], (int)a[1]);
}
// lane 0 receives what `for (off = 32; off; off >>= 1) v += __shfl_down(v, off, 64)` leaves there, bit for bit (other lanes: no meaning)
__device__ __forceinline__ double wave_sum_lane0(double v) {
  v += lanes_plus_32(v);
  v += lanes_plus_16(v);
  v += dpp_row_mov<0x108>(v);      // row_shl:8
  v += dpp_row_mov<0x104>(v);
  v += dpp_row_mov<0x102>(v);
  v += dpp_row_mov<0x101>(v);
  return v;
}
__device__ __forceinline__ double lane_value(double v, int lane_uniform) {       // v of one lane (a wave-uniform index), to all
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane_uniform), __builtin_amdgcn_readlane(__double2loint(v), lane_uniform));
}

#ifndef DFX_DENSE_OCC
#define DFX_DENSE_OCC __attribute__((amdgpu_waves_per_eu(3)))
#endif
namespace {

__device__ __forceinline__ double err_load(const double* p) {
  double v;
  asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void err_store(double* p, double v) {
  asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

template <int MODEL, int CONTACT>
// (no occupancy pin: at three workgroups per compute unit the loop spills 44 B per lane into its stage chain, 31.0 against 27.7 ms for the paper
// workload; 174-182 registers = two workgroups per compute unit = 2 048 waves per launch, wider ensembles follow in further launches)
__global__ __launch_bounds__(kPersistThreads) void k_adaptive_fwd_loop(DevCtx c, AdaptLoopCoef pc, PersistArgs pa, AdaptLoopArgs aa) {
  int ml, w;
  if (!persist_wave(pa, ml, w)) return;
  const int W = pa.waves_per_member;
  const int m = c.m0 + ml;
  Clock ck = c.clock[m];
  if (ck.state) return;
  // the wave stays whole (its reductions are wave-wide): lanes beyond the lattice redo the last block and never store
  LanePos lp = wave_lane_pos<4>(w, c.n_blocks);
  const bool valid = lp.valid;
  if (!valid) { lp.b = c.n_blocks - 1; lp.k = (int)(threadIdx.x & 3); lp.slot = lp.b * 4 + lp.k; }
  const int slot = lp.slot, b = lp.b, k = lp.k, kd = k < 3 ? k : 2;
  const int lane = (int)(threadIdx.x & 63);
  const u32 nd = (u32)c.n_blocks * 3;
  const MemberBases B = member_bases(c, m);
  LigRes g;
  load_lig_res<CONTACT>(c, B, slot, g);
  const int info = g.info, pslot = g.pslot;
  const int dof = b * 3 + kd;
  const u32 o_dof = (u32)dof * 8, o_rec = ((u32)b * kPos + kd) * 8, o_chunk = ((u32)b * kPos + 2 * (k & 1)) * 8;
  const double damp = c.damping_uniform ? B.cst[6 + kd] : ldg<double>(c.damping + (size_t)((u32)m * nd), o_dof);
  const double invm = ldg<double>(c.inv_m + (size_t)((u32)m * nd), o_dof);
  const int sidx = ldg<int>(c.block_special, (u32)b * 4);
  const bool constrained = k < 3 && sidx >= 0 && ((c.special[sidx >= 0 ? sidx : 0].con_mask >> k) & 1);
  const bool dof_lane = valid && k < 3;
  const bool keep = aa.keep != 0;
  double tf_coef[DFX_MAX_FNS];      // k_fwd_persist: a driven / loaded DOF's coefficients of the time functions, resident
#pragma unroll
  for (int f = 0; f < DFX_MAX_FNS; ++f)
    tf_coef[f] = (valid && k < 3 && sidx >= 0 && f < c.n_fns) ? (constrained ? c.special[sidx].con_coef[k][f] : c.special[sidx].load_coef[k][f]) : 0.0;
  double* Am = c.A + (size_t)((u32)m * (u32)(c.s + 1) * nd);
  // ---- the state the launch starts from: y_n (record 0 of step `accepted`, or stage buffer 0) and the FSAL slope A_0
  long long n_acc = ck.accepted;
  double qn, vn;
  BlockRec<double> o;
  {
    const double* P0 = pos_in(c, m, keep ? -1 : 0, n_acc);
    const double* V0 = vel_in(c, m, keep ? -1 : 0, n_acc);
    const double2 a0 = ldg<double2>(P0, (u32)b * kPos * 8), a1 = ldg<double2>(P0, (u32)b * kPos * 8 + 16);
    o.x = a0.x; o.y = a0.y; o.th = a1.x; o.sh = a1.y;
    qn = ldg<double>(P0, o_rec);
    vn = ldg<double>(V0, o_dof);
  }
  double al[7];
  al[0] = ldg<double>(Am, o_dof);
#pragma unroll
  for (int l = 1; l < 7; ++l) al[l] = 0.0;
  // ---- ring addressing (dfx_persist.h) and the partials of the error norm
  const size_t ring_stride = (size_t)c.batch * c.n_blocks * kPos;
  const u32 r_own = ((u32)m * (u32)c.n_blocks + (u32)b) * (kPos * 8) + 16u * (u32)(k & 1);
  const u32 r_par = ((u32)m * (u32)c.n_blocks + (u32)(pslot >> 2)) * (kPos * 8);
  const size_t err_stride = (size_t)c.batch * W;
  double* err_m = aa.err + (size_t)m * W;
  const double* ts_out = c.ts_dev;
  const int Tn = aa.n_timepoints;
  double t = ck.t, h = ck.h;
  double t_out_next = ck.out_idx < Tn ? ts_out[ck.out_idx] : 1.7976931348623157e308;
  int t_ord = 0;
  long long attempt = 0;
  for (; attempt < pa.n_steps; ++attempt) {
    if (keep && n_acc + 3 > aa.cap) break;                         // no room for another kept step: the host grows the buffers
    // ---- the time functions of this attempt at its stage times t + c_j h, j = 0 .. 6: lane j (mod 8) of every wave that holds a driven or
    // loaded block evaluates time j -- the wave pays for ONE evaluation (a software sin / cos) instead of one per stage, and the stages
    // below fetch their values with a lane broadcast.  Same function, same argument as the stage launches: same bits.
    double fg[DFX_MAX_FNS], fgt[DFX_MAX_FNS];
    const bool need_fn = c.n_fns > 0 && __any(sidx >= 0);
    if (need_fn) {
      const int jj = min(lane & 7, 6);
      const double cj = jj == 0 ? pc.c[0] : (jj == 1 ? pc.c[1] : (jj == 2 ? pc.c[2] : (jj == 3 ? pc.c[3] : (jj == 4 ? pc.c[4] : (jj == 5 ? pc.c[5] : pc.c[6])))));
      double gp[kMaxFnParams];
#pragma unroll
      for (int f = 0; f < DFX_MAX_FNS; ++f) {
        fg[f] = 0.0; fgt[f] = 0.0;
        if (f < c.n_fns) eval_time_fn(c.fns[(u32)m * DFX_MAX_FNS + f], t + cj * h, fg[f], fgt[f], gp);
      }
    }
    // value and rate of this lane's prescribed DOF at stage time j (all lanes call: the broadcasts are wave-wide)
    auto prescribed = [&](int j, double& q, double& v) {
      if (!need_fn) return;
      double sq_ = 0.0, sv_ = 0.0;
#pragma unroll
      for (int f = 0; f < DFX_MAX_FNS; ++f) {
        const double gj = lane_value(fg[f], j), gtj = lane_value(fgt[f], j);
        if (constrained && f < c.n_fns) { sq_ += tf_coef[f] * gj; sv_ += tf_coef[f] * gtj; }
      }
      if (constrained) { q = sq_; v = sv_; }
    };
    auto load_at = [&](int j) -> double {
      double fl = 0.0;
      if (!need_fn) return fl;
#pragma unroll
      for (int f = 0; f < DFX_MAX_FNS; ++f) {
        const double gj = lane_value(fg[f], j);
        if (k < 3 && sidx >= 0 && !constrained && f < c.n_fns) fl += tf_coef[f] * gj;
      }
      return fl;
    };
    // ---- record S_1 = y_n + h a_10 k_0 (prescribed DOFs: c(t + c_1 h)), published as ordinal 6 * attempt
    double v_i;
    {
      double qx = qn + h * pc.a[1][0] * vn, vx = vn + h * pc.a[1][0] * al[0];
      prescribed(1, qx, vx);
      v_i = vx;
      const double y1 = blk_bcast<4, 1>(qx, k), th2 = blk_bcast<4, 2>(qx, k), x0 = blk_bcast<4, 0>(qx, k);
      double sn, cs;
      fast_sincos(0.5 * th2, &sn, &cs);
      o.x = x0; o.y = y1; o.th = th2; o.sh = sn;
      if (valid && k < 2) ring_store(pa.ring + (size_t)(t_ord % kPRing) * ring_stride, r_own, k == 0 ? x0 : th2, k == 0 ? y1 : sn);
      if (keep && dof_lane) {
        double* tr = traj_rec(c, m, -2, n_acc);
        if (k < 2) stg<double2>(tr, o_chunk, k == 0 ? make_double2(x0, y1) : make_double2(th2, sn));
        stg<double>(tr + (size_t)c.n_blocks * kPos, o_dof, vx);
      }
    }
    double q1 = 0.0, v1 = 0.0, r2 = 0.0;
#pragma unroll 1
    for (int i = 1; i <= 6; ++i) {
      // ---- in front of the poll: the sums over the earlier slopes for the next record (i == 6: for the error estimate), the load
      o.ch = half_cos(o.th, o.sh);
      double sv = 0.0, sq = 0.0, fload = 0.0;
#pragma unroll
      for (int l = 0; l < 6; ++l) {
        const double a_l = l < i ? al[l] : 0.0;
        sv += (i < 6 ? pc.a[i + 1][l] : pc.e[l]) * a_l;
        sq += (i < 6 ? pc.aa[i + 1][l] : pc.ee[l]) * a_l;
      }
      fload = load_at(i);
      // ---- the partner's record S_i
      double pr[4];
      if (!ring_wait(pa.ring + (size_t)(t_ord % kPRing) * ring_stride, r_par, pr, t_ord, pa.give_up, pa.spin_limit)) return;
      if (valid && k < 2) ring_poison(pa.ring + (size_t)((t_ord + kPAhead) % kPRing) * ring_stride, r_own);
      BlockRec<double> p;
      p.x = pr[0]; p.y = pr[1]; p.th = pr[2]; p.sh = pr[3];
      p.ch = half_cos(p.th, p.sh);
      double fx = 0.0, fy = 0.0, fth = 0.0;
      if (info >= 0) {
        BondGrad<double> bg;
        bond_grad<MODEL, double>(o, p, g.rox, g.roy, g.rpx, g.rpy, g.lx, g.ly, g.l0, g.il0, g.ks, g.ksh, g.kr, g.sgn, bg);
        fx = bg.fx; fy = bg.fy; fth = bg.fth;
        if (CONTACT == 1) {
          const bool far = !(fabs(o.th - p.th) <= g.kap_safe);
          ContactGrad<double> cg;
          contact_grad<double>(g.sgn * (o.th - p.th), far ? g.phi1 : g.phi_min, far ? g.phi2 : g.phi_min, g.am, g.ac, g.kc, cg);
          fth += g.sgn * cg.dkap;
        }
      }
      const double dE = blk_reduce3<4>(fx, fy, fth, k);
      const double a = constrained ? 0.0 : (fload - dE - damp * v_i) * invm;
      ++t_ord;
      if (i < 6) {
        sv += pc.a[i + 1][i] * a;
        sq += pc.aa[i + 1][i] * a;
#pragma unroll
        for (int l = 1; l < 7; ++l) al[l] = l == i ? a : al[l];
        double qx = qn + h * (pc.c[i + 1] * vn + h * sq), vx = vn + h * sv;
        prescribed(i + 1, qx, vx);
        const double y1 = blk_bcast<4, 1>(qx, k), th2 = blk_bcast<4, 2>(qx, k), x0 = blk_bcast<4, 0>(qx, k);
        double sn, cs;
        fast_sincos(0.5 * th2, &sn, &cs);
        o.x = x0; o.y = y1; o.th = th2; o.sh = sn;
        if (valid && k < 2) ring_store(pa.ring + (size_t)(t_ord % kPRing) * ring_stride, r_own, k == 0 ? x0 : th2, k == 0 ? y1 : sn);
        if (keep && dof_lane) {
          double* tr = traj_rec(c, m, -2 - i, n_acc);
          if (k < 2) stg<double2>(tr, o_chunk, k == 0 ? make_double2(x0, y1) : make_double2(th2, sn));
          stg<double>(tr + (size_t)c.n_blocks * kPos, o_dof, vx);
        }
        v_i = vx;
        if (i == 5) { q1 = qx; v1 = vx; }
      } else {
        // the evaluation at the candidate: A_6, and with (e, ee) the embedded error estimate (k_fwd_stage, error mode)
        al[6] = a;
        sv += pc.e[6] * a;
        sq += pc.ee[6] * a;
        if (dof_lane && !constrained) {
          const double eq = h * h * sq, ev = h * sv;
          const double tq = c.atol + c.rtol * fmax(fabs(qn), fabs(q1)), tv = c.atol + c.rtol * fmax(fabs(vn), fabs(v1));
          r2 = (eq / tq) * (eq / tq) + (ev / tv) * (ev / tv);
        }
      }
    }
    // ---- error norm: per-wave sum (the order of k_fwd_stage's error mode), all-gather of the member's partials, k_control's order
    r2 = wave_sum_lane0(r2);
    double* slot_now = err_m + (size_t)(attempt % 3) * err_stride;
    if (lane == 0) err_store(slot_now + w, r2);
    double acc_a = 0.0, acc_b = 0.0;
    for (int r = 0; r * 64 < W; ++r) {
      const int idx = lane + 64 * r;
      double v = 0.0;
      if (idx < W) {
        for (int spins = 0;;) {
          v = err_load(slot_now + idx);
          if (!is_poison(v)) break;
          if (++spins > pa.spin_limit) { *pa.give_up = -1 - (int)(attempt & 0xffff); return; }
          __builtin_amdgcn_s_sleep(1);
        }
      }
      if (r & 1) acc_b += v; else acc_a += v;
    }
    if (lane == 0) err_store(err_m + (size_t)((attempt + 2) % 3) * err_stride + w, __hiloint2double((int)kPoisonWord, (int)kPoisonWord));
    double red = acc_a + acc_b;
    red = lane_value(wave_sum_lane0(red), 0);
    const double ratio = sqrt(red / aa.two_n_free);
    ck.attempts++;
    if (!(ratio == ratio)) { ck.state = 2; break; }
    const double h_new = dopri_next_step(h, ratio);
    if (ratio <= 1.0) {
      const double t_new = t + h;
      const int out_lo = ck.out_idx;
      int out_hi = out_lo;
      if (t_out_next <= t_new) {        // (the next output time rides in a register: no load on the path of a step that crosses none)
        while (out_hi < Tn && ts_out[out_hi] <= t_new) ++out_hi;
        t_out_next = out_hi < Tn ? ts_out[out_hi] : 1.7976931348623157e308;
      }
      if (dof_lane && out_hi > out_lo) {          // dense output for the outputs this step crosses (k_prepare)
        double sm = pc.cm[0] * al[0] + pc.cm[6] * al[6], sma = pc.cma[0] * al[0] + pc.cma[6] * al[6];
#pragma unroll
        for (int l = 1; l < 6; ++l) { sm += pc.cm[l] * al[l]; sma += pc.cma[l] * al[l]; }
        const double qmid = qn + h * (0.5 * vn + h * sma), vmid = vn + h * sm;
        for (int kk = out_lo; kk < out_hi; ++kk) {
          const double tk = ts_out[kk];
          const double r = (tk - t) / (t_new - t);
          double oq = dopri_dense(qn, q1, qmid, vn, v1, h, r), ov = dopri_dense(vn, v1, vmid, al[0], al[6], h, r);
          if (constrained) { const TimeVals tv = constrained_value(c, m, c.special[sidx], k, tk); oq = tv.g; ov = tv.gt; }
          double* f = c.fields_dev + ((size_t)m * Tn + kk) * c.n_blocks * 6;
          stg<double>(f, o_dof, oq);
          stg<double>(f + nd, o_dof, ov);
        }
      }
      if (w == 0 && lane == 0) {                  // what k_control records on accept
        if (c.acc_times && n_acc + 1 <= c.acc_cap) c.acc_times[(size_t)m * c.acc_cap + n_acc] = t_new;
        if (c.step_counts && Tn > 1) c.step_counts[(size_t)m * (Tn - 1) + min(max(out_lo - 1, 0), Tn - 2)]++;
        if (aa.ar.t_steps) {
          double* ts = aa.ar.t_steps + (size_t)m * aa.ar.stride;
          int* op = aa.ar.out_ptr + (size_t)m * aa.ar.stride;
          ts[n_acc + 1] = t_new; ts[n_acc + 2] = t_new;
          op[n_acc] = out_lo; op[n_acc + 1] = out_hi; op[n_acc + 2] = out_hi;
          for (int kk = out_lo; kk < out_hi; ++kk) aa.ar.theta[(size_t)m * Tn + kk] = (ts_out[kk] - t) / (t_new - t);
        }
      }
      // commit: y_n <- y1 (its record is record 0 of the next step already), A_0 <- A_6
      ck.t_last = t; ck.h_acc = h; ck.out_lo = out_lo; ck.out_hi = out_hi; ck.out_idx = out_hi; ck.accepted++;
      t = t_new; ++n_acc;
      qn = q1; vn = v1; al[0] = al[6];
      if (out_hi >= Tn) { ck.state = 1; ck.h = h_new; break; }
    }
    ck.accept = ratio <= 1.0;
    h = h_new;
    if (!(h > 0.0)) { ck.state = 3; break; }
  }
  // ---- what the launch leaves behind: the clock (one lane per member), the FSAL slope, and -- steps not kept -- y_n in stage buffer 0
  ck.t = t; if (ck.state != 1) ck.h = h;
  if (w == 0 && lane == 0) c.clock[m] = ck;
  const double y1 = blk_bcast<4, 1>(qn, k), th2 = blk_bcast<4, 2>(qn, k), x0 = blk_bcast<4, 0>(qn, k);
  double sn, cs;
  fast_sincos(0.5 * th2, &sn, &cs);
  if (dof_lane) {
    stg<double>(Am, o_dof, al[0]);
    if (!keep) {      // the record of y_n, rebuilt from the DOF values
      if (k < 2) stg<double2>(c.POS + (size_t)((u32)m * (u32)c.nbuf * (u32)c.n_blocks * kPos), o_chunk, k == 0 ? make_double2(x0, y1) : make_double2(th2, sn));
      stg<double>(c.VEL + (size_t)((u32)m * (u32)c.nbuf * nd), o_dof, vn);
    }
  }
}

// the reverse sweep of the kept steps without kernel boundaries: adj_persist_body<..., DENSE = 1> (dfx_persist.h)
template <int MODEL, int CONTACT, int NPB>
__global__ __launch_bounds__(kPersistThreads) DFX_DENSE_OCC void k_adj_dense_loop(DevCtx c, PersistAdjCoef pc, PersistArgs pa,
                                                                                                           DenseCtx dn) {
  adj_persist_body<MODEL, CONTACT, NPB, 1>(c, pc, pa, dn);
}

}  // namespace
