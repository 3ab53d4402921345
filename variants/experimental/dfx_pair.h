// dfx_pair.h -- two Runge-Kutta stages per launch on overlapping lattice windows (gfx950).  Included by dfx_engine.hip after
// dfx_kernels.h.
//
// Why: both stage kernels move about their algorithmic bytes at ~80 % of the rate this chip streams; what is left per step is the
// state one launch per stage hands to the next THROUGH MEMORY -- stage records, w, Ybar, the accumulators' read-modify-write, the
// parameters each launch re-reads -- and the launch count.  A pair launch evaluates stage i on a WINDOW of the lattice (16 columns x
// WROWS rows of blocks, one wavefront per window row: 16 consecutive blocks = 64 lanes, so every per-block array is still read in
// contiguous runs), keeps what stage i+1 gathers from its neighbours (forward: the new stage records; reverse: w = Kbar_v / m) in LDS,
// and evaluates the second stage on the window's interior from there: the second stage reads no records / no w, the parameters and the
// per-DOF history stay in registers, the gradient accumulators are read and written once per pair, the launch count halves.  The
// window's outer ring is evaluated twice (by this window and by the neighbour that owns it): 14 x 14 of 16 x 16 blocks are owned.
//
// Windows need no permutation of the caller's block order: the host only has to find a row length R with block = row * R + col such
// that every ligament joins blocks at most one row and one column apart (dfx_engine.hip: find_tiling; quads: R = n1_blocks, kagome:
// R = 2 n1_cells).  Connectivity without such an R (or one single row) keeps the one-stage launches.
//
//   k_fwd_pair   stages (i, i+1), i even
//   k_adj_pair   reverse stages (i, i-1), i odd; records checkpoint only (the reverse launch reads the records it linearises about)
#pragma once
#include "dfx_kernels.h"

namespace {

struct TileCtx {
  int R, n_rows;          // block = row * R + col
  int tiles_x, tiles_y, n_tiles;
  int total_wg;           // n_tiles * members of the group
};

constexpr int kWCols = 16;   // blocks per window row = lanes per wave / 4
#ifndef DFX_PAIR_OCC
#define DFX_PAIR_OCC
#endif

// Window of this workgroup and the place of this lane in it.  Workgroups are dealt to the XCDs round-robin: logical_wg gives every
// XCD a contiguous run of (member, tile) pairs, so that neighbouring windows -- which read each other's outer ring -- share an L2.
template <int WROWS>
struct WinPos {
  int m, b, widx, row, ws_x, ws_y;
  bool valid, owned;
};
template <int WROWS>
__device__ __forceinline__ WinPos<WROWS> window_pos(const DevCtx& c, const TileCtx& tc) {
  WinPos<WROWS> w;
  const int lw = logical_wg(blockIdx.x, tc.total_wg);
  const int ml = lw / tc.n_tiles, tile = lw - ml * tc.n_tiles;
  w.m = c.m0 + ml;
  const int ty = tile / tc.tiles_x, tx = tile - ty * tc.tiles_x;
  w.ws_x = tx * (kWCols - 2);
  w.ws_y = ty * (WROWS - 2);
  const int wv = threadIdx.x >> 6, cl = (threadIdx.x & 63) >> 2;
  w.row = w.ws_y + wv;
  const int col = w.ws_x + cl;
  w.valid = w.row < tc.n_rows && col < tc.R;
  // owned: everything but the window's outer ring; at a lattice edge the ring is the edge and is owned too
  const int ox0 = tx == 0 ? 0 : w.ws_x + 1, ox1 = w.ws_x + kWCols >= tc.R ? tc.R : w.ws_x + kWCols - 1;
  const int oy0 = ty == 0 ? 0 : w.ws_y + 1, oy1 = w.ws_y + WROWS >= tc.n_rows ? tc.n_rows : w.ws_y + WROWS - 1;
  w.owned = w.valid && col >= ox0 && col < ox1 && w.row >= oy0 && w.row < oy1;
  w.b = w.row * tc.R + col;
  w.widx = wv * kWCols + cl;
  return w;
}
// place of a partner block (at most one row / column away) in the window
template <int WROWS>
__device__ __forceinline__ int window_index(const WinPos<WROWS>& w, const TileCtx& tc, int pb) {
  const int r0 = w.row * tc.R;
  const int prow = w.row + (pb >= r0 + tc.R ? 1 : 0) - (pb < r0 ? 1 : 0);
  return (prow - w.ws_y) * kWCols + (pb - prow * tc.R - w.ws_x);
}

// the step state ping-pongs between stage buffers 0 and 3 in the pair launches (a window reads q_n, v_n of its outer ring, which a
// neighbouring window may already have advanced if the new state went to the same buffer)
constexpr int kStateBufOdd = 3;
__host__ __device__ __forceinline__ int state_buf(long long n) { return (n & 1) ? kStateBufOdd : 0; }

// ---- forward pair --------------------------------------------------------------------------------------------------------------
//   in_buf   records of stage i: stage buffer (0 = the step state: buffer state_buf(n)), or < 0: record (-1 - in_buf) of step n in
//            the checkpoint
//   mid_buf  where the record of stage i+1 goes: -1 nowhere (nobody reads it after this launch), < -1 checkpoint record
//   out_buf  record of stage i+2: stage buffer (0 = the state of step n+1: buffer state_buf(n+1)), < -1 checkpoint record
//   y_buf    0: (q_n, v_n) in buffer state_buf(n);  -1: in the checkpoint of step n
//   mode & 1 also store the new step state into the checkpoint of step n+1 (last pair, state / stages level)
template <int MODEL, int CONTACT, int WROWS>
__global__ __launch_bounds__(64 * WROWS) DFX_PAIR_OCC void k_fwd_pair(DevCtx c, TileCtx tc, StageCoef sc0, StageCoef sc1, int i, int j, int in_buf,
                                                         int mid_buf, int out_buf, int y_buf, int mode) {
  __shared__ double2 s_rec[WROWS * kWCols * 2];
  const WinPos<WROWS> w = window_pos<WROWS>(c, tc);
  const int m = w.m, b = w.b, k = threadIdx.x & 3, kd = k < 3 ? k : 2;
  const int slot = b * 4 + k;
  const Seg sg = *c.cur;
  const long long n = sg.base_step + j;
  if (in_buf == 0) in_buf = state_buf(n);
  if (out_buf == 0) out_buf = state_buf(n + 1);
  const int ys = y_buf == 0 ? state_buf(n) : y_buf;
  const u32 nd = (u32)c.n_blocks * 3;
  const int dof = b * 3 + kd;
  const u32 o_dof = (u32)dof * 8, o_rec = ((u32)b * kPos + kd) * 8;
  const u32 o_chunk = ((u32)b * kPos + 2 * k) * 8;
  const MemberBases B = member_bases(c, m);
  const bool keep_stages = c.AD != nullptr;
  double* Am = keep_stages ? c.AD + (size_t)m * c.ad_stride + (size_t)n * ((u32)(c.s - 1) * nd) : c.A + (size_t)m * (u32)(c.s + 1) * nd;
  double h = sg.h, t = sg.t_interval + (sg.j0 + j) * sg.h;
  if (c.t_steps) { const double* ts = steps_of(c, m); t = ts[n]; h = ts[n + 1] - t; }
  // what survives the barrier
  LaneIn L;
  double qn = 0.0, vn = 0.0, damp = 0.0, invm = 0.0, qnext = 0.0, vnext = 0.0, sv1 = 0.0, sq1 = 0.0, sn = 0.0;
  int sidx = -1;
  L.info = -1; L.pslot = 0;
  if (w.valid) {
    // ---- stage i on the whole window: the load phase of k_fwd_stage
    const double* POSin = pos_in(c, m, in_buf, n);
    LaneRaw R;
    issue_lane<CONTACT>(c, B, slot, POSin, R);
    qn = ldg<double>(pos_in(c, m, ys, n), o_rec);
    vn = ldg<double>(vel_in(c, m, ys, n), o_dof);
    const double v_i = ldg<double>(vel_in(c, m, in_buf, n), o_dof);
    damp = c.damping_uniform ? B.cst[6 + kd] : ldg<double>(c.damping + (size_t)m * nd, o_dof);
    invm = ldg<double>(c.inv_m + (size_t)m * nd, o_dof);
    sidx = ldg<int>(c.block_special, (u32)b * 4);
    double al[kMaxStages - 1];
#pragma unroll
    for (int l = 0; l < kMaxStages - 1; ++l) al[l] = l < i ? ldg<double>(Am + (size_t)l * nd, o_dof) : 0.0;
    resolve_lane<CONTACT, 4, WROWS>(c, B, POSin, R, L);
    double sv = 0.0, sq = 0.0;
#pragma unroll
    for (int l = 0; l < kMaxStages - 1; ++l) {
      sv += sc0.cv[l] * al[l];
      sq += sc0.cq[l] * al[l];
      sv1 += sc1.cv[l] * al[l];
      sq1 += sc1.cq[l] * al[l];
    }
    double fx = 0.0, fy = 0.0, fth = 0.0;
    if (L.info >= 0) {
      BondGrad<double> g;
      bond_grad<MODEL, double>(L.o, L.p, L.rox, L.roy, L.rpx, L.rpy, L.lx, L.ly, L.l0, L.il0, L.ks, L.ksh, L.kr, L.sgn, g);
      fx = g.fx; fy = g.fy; fth = g.fth;
      if (CONTACT == 1) {
        ContactGrad<double> cg;
        contact_grad<double>(L.sgn * (L.o.th - L.p.th), L.phi1, L.phi2, L.am, L.ac, L.kc, cg);
        fth += L.sgn * cg.dkap;
      }
    }
    fx = quad_sum(fx);
    fy = quad_sum(fy);
    fth = quad_sum(fth);
    if (k < 3) {
      const double dE = k == 0 ? fx : (k == 1 ? fy : fth);
      bool constrained = false;
      double fload = 0.0;
      if (sidx >= 0) {
        const dfx_special& sp = c.special[sidx];
        constrained = (sp.con_mask >> k) & 1;
        if (!constrained) {
          double gp[kMaxFnParams];
          for (int f = 0; f < c.n_fns; ++f)
            if (sp.load_coef[k][f] != 0.0) {
              double g, gt;
              eval_time_fn(c.fns[(u32)m * DFX_MAX_FNS + f], t + sc0.c_i * h, g, gt, gp);
              fload += sp.load_coef[k][f] * g;
            }
        }
      }
      const double a = constrained ? 0.0 : (fload - dE - damp * v_i) * invm;
      if (w.owned) stg<double>(Am + (size_t)i * nd, o_dof, a);
      sv += sc0.cv[i] * a;
      sq += sc0.cq[i] * a;
      sv1 += sc1.cv[i] * a;
      sq1 += sc1.cq[i] * a;
      qnext = qn + h * (sc0.c_next * vn + h * sq);
      vnext = vn + h * sv;
      if (constrained) {
        TimeVals tv = constrained_value(c, m, c.special[sidx], k, t + sc0.c_next * h);
        qnext = tv.g; vnext = tv.gt;
      }
    }
    // the record of stage i+1: into LDS for the neighbours, into the checkpoint when the reverse sweep will read it
    const double y1 = quad_bcast<1>(qnext), th2 = quad_bcast<2>(qnext);
    double cs;
    fast_sincos(0.5 * th2, &sn, &cs);
    const double2 chunk = k == 0 ? make_double2(qnext, y1) : make_double2(th2, sn);
    if (k < 2) s_rec[w.widx * 2 + k] = chunk;
    if (w.owned && mid_buf < -1 && k < 3) {
      double* tr = traj_rec(c, m, mid_buf, n);
      if (k < 2) stg<double2>(tr, o_chunk, chunk);
      stg<double>(tr + (size_t)c.n_blocks * kPos, o_dof, vnext);
    }
  }
  __syncthreads();
  if (!w.owned) return;
  // ---- stage i+1 on the owned blocks: own record from registers, the partner's from LDS, parameters as loaded above
  {
    L.o.x = quad_bcast<0>(qnext); L.o.y = quad_bcast<1>(qnext); L.o.th = quad_bcast<2>(qnext); L.o.sh = sn;
    L.o.ch = half_cos(L.o.th, L.o.sh);
    const int pidx = L.info >= 0 ? window_index<WROWS>(w, tc, L.pslot >> 2) : w.widx;
    const double2 p0 = s_rec[pidx * 2], p1 = s_rec[pidx * 2 + 1];
    L.p.x = p0.x; L.p.y = p0.y; L.p.th = p1.x; L.p.sh = p1.y;
    L.p.ch = half_cos(L.p.th, L.p.sh);
    if (CONTACT == 1) {
      const double* cst = B.cst;
      double2 ph = make_double2(cst[10], cst[10]);
      if (L.info >= 0 && !(fabs(L.o.th - L.p.th) <= cst[9])) ph = ldg<double2>(B.p_phi, (u32)slot * 16);
      L.phi1 = ph.x;
      L.phi2 = ph.y;
    }
    double fx = 0.0, fy = 0.0, fth = 0.0;
    if (L.info >= 0) {
      BondGrad<double> g;
      bond_grad<MODEL, double>(L.o, L.p, L.rox, L.roy, L.rpx, L.rpy, L.lx, L.ly, L.l0, L.il0, L.ks, L.ksh, L.kr, L.sgn, g);
      fx = g.fx; fy = g.fy; fth = g.fth;
      if (CONTACT == 1) {
        ContactGrad<double> cg;
        contact_grad<double>(L.sgn * (L.o.th - L.p.th), L.phi1, L.phi2, L.am, L.ac, L.kc, cg);
        fth += L.sgn * cg.dkap;
      }
    }
    fx = quad_sum(fx);
    fy = quad_sum(fy);
    fth = quad_sum(fth);
    double q2 = 0.0, v2 = 0.0;
    const int i1 = i + 1;
    if (k < 3) {
      const double dE = k == 0 ? fx : (k == 1 ? fy : fth);
      bool constrained = false;
      double fload = 0.0;
      if (sidx >= 0) {
        const dfx_special& sp = c.special[sidx];
        constrained = (sp.con_mask >> k) & 1;
        if (!constrained) {
          double gp[kMaxFnParams];
          for (int f = 0; f < c.n_fns; ++f)
            if (sp.load_coef[k][f] != 0.0) {
              double g, gt;
              eval_time_fn(c.fns[(u32)m * DFX_MAX_FNS + f], t + sc1.c_i * h, g, gt, gp);
              fload += sp.load_coef[k][f] * g;
            }
        }
      }
      const double a = constrained ? 0.0 : (fload - dE - damp * vnext) * invm;
      if (!(keep_stages && i1 == c.s - 1)) stg<double>(Am + (size_t)i1 * nd, o_dof, a);
      sv1 += sc1.cv[i1] * a;
      sq1 += sc1.cq[i1] * a;
      q2 = qn + h * (sc1.c_next * vn + h * sq1);
      v2 = vn + h * sv1;
      if (constrained) {
        TimeVals tv = constrained_value(c, m, c.special[sidx], k, t + sc1.c_next * h);
        q2 = tv.g; v2 = tv.gt;
      }
    }
    const double y1 = quad_bcast<1>(q2), th2 = quad_bcast<2>(q2);
    double sn2, cs2;
    fast_sincos(0.5 * th2, &sn2, &cs2);
    const double2 chunk = k == 0 ? make_double2(q2, y1) : make_double2(th2, sn2);
    if (k < 3) {
      if (out_buf >= 0) {
        if (k < 2) stg<double2>(c.POS + ((size_t)m * c.nbuf + out_buf) * (u32)c.n_blocks * kPos, o_chunk, chunk);
        stg<double>(c.VEL + ((size_t)m * c.nbuf + out_buf) * nd, o_dof, v2);
      }
      if ((mode & 1) || out_buf < -1) {
        double* tr = out_buf < -1 ? traj_rec(c, m, out_buf, n) : traj_rec(c, m, -1, n + 1);
        if (k < 2) stg<double2>(tr, o_chunk, chunk);
        stg<double>(tr + (size_t)c.n_blocks * kPos, o_dof, v2);
      }
    }
  }
}

// ---- reverse pair ----------------------------------------------------------------------------------------------------------------
// Reverse stages i (odd; on the whole window) and i-1 (on the owned blocks) of step n, records checkpoint.  Between the two:
// w_{i-1} = Kbar_v,{i-1} / m of every window block goes through LDS.  lambda is double-buffered by step parity (a window reads the
// lambda_{n+1} of its outer ring while the neighbour that owns the ring may already have written lambda_n), (w) by launch parity.
//   ac1 = adj_coef(i), ac2 = adj_coef(i - 1)
template <int MODEL, int CONTACT, int WROWS>
__global__ __launch_bounds__(64 * WROWS) DFX_PAIR_OCC void k_adj_pair(DevCtx c, TileCtx tc, AdjCoef ac1, AdjCoef ac2, int i, int j) {
  __shared__ double s_w[WROWS * kWCols * 3];
  const WinPos<WROWS> w = window_pos<WROWS>(c, tc);
  const int m = w.m, b = w.b, k = threadIdx.x & 3, kd = k < 3 ? k : 2;
  const int slot = b * 4 + k;
  const Seg sg = *c.cur;
  const long long n = sg.base_step + j;
  const int half = c.s >> 1;
  const int win = (int)((n * half + (i >> 1)) & 1);
  const u32 nd = (u32)c.n_blocks * 3, nd6 = (u32)c.n_blocks * 6;
  const int dof = b * 3 + kd;
  const u32 o_dof = (u32)dof * 8, o_b6 = ((u32)b * 6 + 2 * kd) * 8;
  const MemberBases B = member_bases(c, m);
  double h = sg.h, t_n = sg.t_interval + (sg.j0 + j) * sg.h, h_before = (sg.j0 + j) == 0 ? sg.h_prev : sg.h;
  if (c.t_steps) { const double* ts = steps_of(c, m); t_n = ts[n]; h = ts[n + 1] - t_n; h_before = n > 0 ? t_n - ts[n - 1] : 0.0; }
  double* YBm = c.YB + (size_t)m * (u32)c.s * nd6;
  const double* LAMin = c.LAM + ((size_t)((n + 1) & 1) * (u32)c.batch + (u32)m) * nd6;
  double* LAMout = c.LAM + ((size_t)(n & 1) * (u32)c.batch + (u32)m) * nd6;
  const double* Win = c.W + ((size_t)m * 2 + win) * nd;
  // what survives the barrier
  LaneIn L;
  L.info = -1; L.pslot = 0;
  double damp = 0.0, invm = 0.0, lq = 0.0, lv = 0.0, sq2 = 0.0, sv2 = 0.0, kq1 = 0.0, w1 = 0.0;
  double d_rx = 0.0, d_ry = 0.0, d_phi = 0.0, d_bm = 0.0, d_bc = 0.0;
  int sidx = -1;
  bool constrained = false;
  if (w.valid) {
    const double* POSin = traj_rec(c, m, -1 - i, n);
    LaneRaw R;
    issue_lane<CONTACT>(c, B, slot, POSin, R);
    double wpx, wpy, wpth;
    { const u32 gb = (u32)(R.guess >> 2) * 24;
      const double2 wxy = ldg<double2>(Win, gb); wpx = wxy.x; wpy = wxy.y;
      wpth = ldg<double>(Win, gb + 16); }
    const double v_i = w.owned ? ldg<double>(POSin + (size_t)c.n_blocks * kPos, o_dof) : 0.0;
    damp = c.damping_uniform ? B.cst[6 + kd] : ldg<double>(c.damping + (size_t)m * nd, o_dof);
    invm = ldg<double>(c.inv_m + (size_t)m * nd, o_dof);
    sidx = ldg<int>(c.block_special, (u32)b * 4);
    { const double2 l2 = ldg<double2>(LAMin, o_b6); lq = l2.x; lv = l2.y; }
    double2 yb[kMaxStages];
#pragma unroll
    for (int jj = 1; jj < kMaxStages; ++jj) {
      const bool on = jj > i && jj < c.s;
      yb[jj] = on ? ldg<double2>(YBm + (size_t)jj * nd6, o_b6) : make_double2(0.0, 0.0);
    }
    double sq = 0.0, sv = 0.0, sqc = 0.0, svc = 0.0;
#pragma unroll
    for (int jj = 1; jj < kMaxStages; ++jj) {
      sq += ac1.col[jj] * yb[jj].x;
      sv += ac1.col[jj] * yb[jj].y;
      sqc += ac1.cur[jj] * yb[jj].x;
      svc += ac1.cur[jj] * yb[jj].y;
      const double c2 = i > 1 ? ac2.col[jj] : 1.0;      // second stage = stage 0: lambda_n = lambda_{n+1} + sum_j Ybar_j
      sq2 += c2 * yb[jj].x;
      sv2 += c2 * yb[jj].y;
    }
    const double w_d = (h * (ac1.cur[c.s] * lv + svc)) * invm;
    resolve_lane<CONTACT, 4, WROWS>(c, B, POSin, R, L);
    if (L.pslot != L.guess) {
      const u32 pb = (u32)(L.pslot >> 2) * 24;
      const double2 wxy = ldg<double2>(Win, pb); wpx = wxy.x; wpy = wxy.y;
      wpth = ldg<double>(Win, pb + 16);
    }
    const double wox = quad_bcast<0>(w_d), woy = quad_bcast<1>(w_d), woth = quad_bcast<2>(w_d);
    double hx = 0.0, hy = 0.0, hth = 0.0, ex = 0.0, ey = 0.0, eth = 0.0;
    if (L.info >= 0) {
      BlockRec<Dual> o = seed_rec(L.o, wox, woy, woth);
      BlockRec<Dual> p = seed_rec(L.p, wpx, wpy, wpth);
      BondGrad<Dual> g;
      bond_grad<MODEL, Dual>(o, p, L.rox, L.roy, L.rpx, L.rpy, L.lx, L.ly, L.l0, L.il0, L.ks, L.ksh, L.kr, L.sgn, g);
      hx = g.fx.e; hy = g.fy.e; hth = g.fth.e;
      ex = g.fx.v; ey = g.fy.v; eth = g.fth.v;
      d_rx = g.rx.e; d_ry = g.ry.e;
      if (CONTACT == 1) {
        ContactGrad<Dual> cg;
        contact_grad<Dual>(L.sgn * (o.th - p.th), L.phi1, L.phi2, L.am, L.ac, L.kc, cg);
        hth += L.sgn * cg.dkap.e;
        eth += L.sgn * cg.dkap.v;
        d_phi = (L.info & 1) ? cg.p2.e : cg.p1.e;
      }
    }
    hx = quad_sum(hx);
    hy = quad_sum(hy);
    hth = quad_sum(hth);
    ex = quad_sum(ex);
    ey = quad_sum(ey);
    eth = quad_sum(eth);
    if (k < 3) {
      const double hw = k == 0 ? hx : (k == 1 ? hy : hth);
      const double dE = k == 0 ? ex : (k == 1 ? ey : eth);
      double fload = 0.0;
      if (sidx >= 0) {
        const dfx_special& sp = c.special[sidx];
        constrained = (sp.con_mask >> k) & 1;
        const double t_i = t_n + ac1.c_i * h;
        double gp[kMaxFnParams];
        for (int f = 0; f < c.n_fns; ++f) {
          const double coef = constrained ? -hw * sp.con_coef[k][f] : w_d * sp.load_coef[k][f];
          const bool loaded = !constrained && sp.load_coef[k][f] != 0.0;
          const bool grad = coef != 0.0 && c.fn_g && w.owned;
          if (grad || loaded) {
            double g, gt;
            eval_time_fn(c.fns[(u32)m * DFX_MAX_FNS + f], t_i, g, gt, gp);
            if (loaded) fload += sp.load_coef[k][f] * g;
            if (grad) {
              double* q = c.fn_g + (((size_t)m * c.n_special + sidx) * DFX_MAX_FNS + f) * DFX_FN_PARAMS;
              for (int kk = 0; kk < DFX_FN_PARAMS; ++kk) acc_add(q + kk, coef * gp[kk]);
            }
          }
        }
      }
      const double a_i = constrained ? 0.0 : (fload - dE - damp * v_i) * invm;
      const double kq_in = h * (ac1.cur[c.s] * lq + sqc);
      double ybq = 0.0, ybv = 0.0;
      if (!constrained) {
        ybq = -hw;
        ybv = kq_in - damp * w_d;
        d_bm = w_d * a_i;
        d_bc = w_d * v_i;
      }
      if (w.owned) stg<double2>(YBm + (size_t)i * nd6, o_b6, make_double2(ybq, ybv));
      kq1 = h * (ac1.col[c.s] * lq + ac1.col[i] * ybq + sq);
      const double kv1 = h * (ac1.col[c.s] * lv + ac1.col[i] * ybv + sv);
      w1 = constrained ? 0.0 : kv1 * invm;
      const double c2 = i > 1 ? ac2.col[i] : 1.0;
      sq2 += c2 * ybq;
      sv2 += c2 * ybv;
      s_w[w.widx * 3 + k] = w1;
    }
  }
  __syncthreads();
  if (!w.owned) return;
  // ---- stage i-1 on the owned blocks
  {
    const int i0 = i - 1;
    const double* POS2 = traj_rec(c, m, -1 - i0, n);
    const double2 pc = k < 2 ? ldg<double2>(POS2, ((u32)b * kPos + 2 * k) * 8) : make_double2(0.0, 0.0);
    const u32 prec = (u32)(L.pslot >> 2) * (kPos * 8);
    const double2 pb0 = ldg<double2>(POS2, prec), pb1 = ldg<double2>(POS2, prec + 16);
    const double v_i = ldg<double>(POS2 + (size_t)c.n_blocks * kPos, o_dof);
    // the accumulators' old values: one batch with the record
    const size_t ms = (size_t)m * (u32)c.n_slots;
    double* grm = c.g_r + ms * 2;
    double* gpm = c.g_phi + ms;
    double* bmm = c.blk_m + (size_t)m * nd;
    double* bcm = c.blk_c + (size_t)m * nd;
    const double2 r_old = ldg<double2>(grm, (u32)slot * 16);
    const double bm_old = ldg<double>(bmm, o_dof);
    const double bc_old = c.blk_c ? ldg<double>(bcm, o_dof) : 0.0;
    const int pidx = L.info >= 0 ? window_index<WROWS>(w, tc, L.pslot >> 2) : w.widx;
    const double wpx = s_w[pidx * 3], wpy = s_w[pidx * 3 + 1], wpth = s_w[pidx * 3 + 2];
    const double wox = quad_bcast<0>(w1), woy = quad_bcast<1>(w1), woth = quad_bcast<2>(w1);
    L.o.x = quad_bcast<0>(pc.x); L.o.y = quad_bcast<0>(pc.y);
    L.o.th = quad_bcast<1>(pc.x); L.o.sh = quad_bcast<1>(pc.y);
    L.o.ch = half_cos(L.o.th, L.o.sh);
    L.p.x = pb0.x; L.p.y = pb0.y; L.p.th = pb1.x; L.p.sh = pb1.y;
    L.p.ch = half_cos(L.p.th, L.p.sh);
    if (CONTACT == 1) {
      const double* cst = B.cst;
      double2 ph = make_double2(cst[10], cst[10]);
      if (L.info >= 0 && !(fabs(L.o.th - L.p.th) <= cst[9])) ph = ldg<double2>(B.p_phi, (u32)slot * 16);
      L.phi1 = ph.x;
      L.phi2 = ph.y;
    }
    double hx = 0.0, hy = 0.0, hth = 0.0, ex = 0.0, ey = 0.0, eth = 0.0;
    if (L.info >= 0) {
      BlockRec<Dual> o = seed_rec(L.o, wox, woy, woth);
      BlockRec<Dual> p = seed_rec(L.p, wpx, wpy, wpth);
      BondGrad<Dual> g;
      bond_grad<MODEL, Dual>(o, p, L.rox, L.roy, L.rpx, L.rpy, L.lx, L.ly, L.l0, L.il0, L.ks, L.ksh, L.kr, L.sgn, g);
      hx = g.fx.e; hy = g.fy.e; hth = g.fth.e;
      ex = g.fx.v; ey = g.fy.v; eth = g.fth.v;
      d_rx += g.rx.e; d_ry += g.ry.e;
      if (CONTACT == 1) {
        ContactGrad<Dual> cg;
        contact_grad<Dual>(L.sgn * (o.th - p.th), L.phi1, L.phi2, L.am, L.ac, L.kc, cg);
        hth += L.sgn * cg.dkap.e;
        eth += L.sgn * cg.dkap.v;
        d_phi += (L.info & 1) ? cg.p2.e : cg.p1.e;
      }
    }
    const bool phi_on = CONTACT == 1 && d_phi != 0.0;
    const double p_old = phi_on ? ldg<double>(gpm, (u32)slot * 8) : 0.0;
    hx = quad_sum(hx);
    hy = quad_sum(hy);
    hth = quad_sum(hth);
    ex = quad_sum(ex);
    ey = quad_sum(ey);
    eth = quad_sum(eth);
    if (L.info >= 0) stg<double2>(grm, (u32)slot * 16, make_double2(r_old.x - d_rx, r_old.y - d_ry));
    if (phi_on) { stg<double>(gpm, (u32)slot * 8, p_old - d_phi); c.touch[0] = 1; }
    if (k < 3) {
      const double hw = k == 0 ? hx : (k == 1 ? hy : hth);
      const double dE = k == 0 ? ex : (k == 1 ? ey : eth);
      double fload = 0.0;
      if (sidx >= 0) {
        const dfx_special& sp = c.special[sidx];
        const double t_i = t_n + ac2.c_i * h;
        double gp[kMaxFnParams];
        for (int f = 0; f < c.n_fns; ++f) {
          const double coef = constrained ? -hw * sp.con_coef[k][f] : w1 * sp.load_coef[k][f];
          const bool loaded = !constrained && sp.load_coef[k][f] != 0.0;
          if ((coef != 0.0 && c.fn_g) || loaded) {
            double g, gt;
            eval_time_fn(c.fns[(u32)m * DFX_MAX_FNS + f], t_i, g, gt, gp);
            if (loaded) fload += sp.load_coef[k][f] * g;
            if (coef != 0.0 && c.fn_g) {
              double* q = c.fn_g + (((size_t)m * c.n_special + sidx) * DFX_MAX_FNS + f) * DFX_FN_PARAMS;
              for (int kk = 0; kk < DFX_FN_PARAMS; ++kk) acc_add(q + kk, coef * gp[kk]);
            }
          }
        }
      }
      const double a_i = constrained ? 0.0 : (fload - dE - damp * v_i) * invm;
      double ybq = 0.0, ybv = 0.0;
      if (!constrained) {
        ybq = -hw;
        ybv = kq1 - damp * w1;
        d_bm += w1 * a_i;
        d_bc += w1 * v_i;
        stg<double>(bmm, o_dof, bm_old - d_bm);
        if (c.blk_c) stg<double>(bcm, o_dof, bc_old - d_bc);
      }
      stg<double2>(YBm + (size_t)i0 * nd6, o_b6, make_double2(ybq, ybv));
      double kv;
      if (i0 > 0) {
        kv = h * (ac2.col[c.s] * lv + ac2.col[i0] * ybv + sv2);
      } else {
        lq += ybq + sq2;
        lv += ybv + sv2;
        const bool first = (sg.j0 + j) == 0;
        if (first && c.G && !constrained) {
          const double* G = c.G + ((size_t)sg.interval * c.batch + m) * (size_t)nd6;
          lq += G[b * 6 + k]; lv += G[b * 6 + 3 + k];
        }
        if (constrained) { lq = 0.0; lv = 0.0; }
        stg<double2>(LAMout, o_b6, make_double2(lq, lv));
        kv = h_before * ac2.col[c.s] * lv;
      }
      stg<double>(c.W + ((size_t)m * 2 + (win ^ 1)) * nd, o_dof, constrained ? 0.0 : kv * invm);
    }
  }
}

}  // namespace
