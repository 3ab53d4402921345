// dfx_tile.h -- the two stage kernels with every ligament evaluated ONCE, on lattice tiles (round 4).
// Included by dfx_engine.hip after dfx_kernels.h.
//
// Why.  k_fwd_stage / k_adj_stage map one lane to one (block, node slot): every ligament is evaluated by both of its end lanes, and
// each lane gathers its partner's record (and w) from memory.  The counters of round 3 say those launches keep the vector ALUs busy
// half of the time on top of a memory system at 60 %, and that the gathers cost 6 us of a 33 us reverse launch.  The reference
// evaluates a bond once (jax_md.smap.bond over the bond list, energy.py:179-197); so do these kernels:
//
//   * a workgroup of NW wavefronts owns a tile of 16 x TH blocks (TH = 2 NW - 1 lattice rows of a grid found by find_tiling:
//     block = row * R + col).  Two lanes per block: lane e of block b evaluates the ligament the block OWNS in direction e
//     (e = 0: partner b + 1; e = 1: partner b + R + dc1, dc1 one of -1, 0, +1 per lattice) and is the DOF lane of (x, y) (e = 0) or
//     theta (e = 1) in the integrator epilogue -- 32 blocks per wave instead of 16;
//   * the 32 lanes left over in the last wave ("crew") load the ring of blocks around the tile into LDS and evaluate the ligaments
//     that enter the tile from outside (owned by a block of another tile; at most 2 TH + 15): those are the only ligaments
//     evaluated twice (23 of 247 per tile of 16 x 7);
//   * stage records (and, reverse, w = Kbar_v / m) of the tile + ring live in LDS: no gathers from memory at all;
//   * one evaluation yields both ends' derivatives (bond_grad with BondPartner); the partner's half goes through LDS to the DOF
//     lanes of the partner block, summed in a fixed order (own e = 0, own e = 1, from the left, from below): bit-reproducible;
//   * parameters and gradient accumulators are ligament-major (LigCtx: both node vectors of a ligament side by side, d/d(node
//     vectors) of both ends in one 32-byte accumulator), built from / folded back into the slot-major arrays by k_lig_pack /
//     k_lig_unpack once per set_params / per sweep, so nothing outside this file changes its layout.
//
// Covered: 4- and 3-node blocks on a grid, one ligament per node, no distance-based contact, fixed grid with the segment's
// time-function table (or no time function), reverse: the build without per-ligament gradients and without the stage rebuild.
// Everything else keeps the slot kernels (lig_fwd_ok / lig_adj_ok in dfx_engine.hip).
#pragma once
#include "dfx_kernels.h"

namespace {

constexpr int kTileW = 16;          // blocks per tile row: 16 x 32 B = four full cache lines of records per row and wave half

struct LigCtx {
  int R, n_rows, dc1, tiles_x;      // grid of the lattice: block = row * R + col; column offset of the e = 1 partner
  int n_wg, pad0;                   // tiles per member
  const int32_t* tab;               // batch * 2n  bit 0: the lane has a ligament, bit 1: the owner is end 2 of the bond, bits 8..15: dictionary index
  const double* p;                  // batch * 2n * 4   node vector of the owner's node, node vector of the partner's node
  const double* l;                  // batch * 2n * 2   reference vector (read when the dictionary is not in LDS)
  const double* k;                  // batch * 2n * 4   stiffnesses (read when they differ between ligaments)
  const double* phi;                // batch * 2n * 2   undeformed void angles (phi1, phi2)
  double* g;                        // batch * 2n * 4   d/d(owner's node vector), d/d(partner's node vector)
  double* gphi;                     // batch * 2n * 2   d/d(phi1, phi2)
};

typedef double dfx_d2u __attribute__((ext_vector_type(2), aligned(8)));     // two doubles at an 8-byte aligned address (per-DOF arrays: 24 B per block)
__device__ __forceinline__ double2 ldu2(const void* base, u32 byte_off) {
  const dfx_d2u v = *reinterpret_cast<const dfx_d2u*>(reinterpret_cast<const char*>(base) + byte_off);
  return make_double2(v.x, v.y);
}
__device__ __forceinline__ void stu2(void* base, u32 byte_off, double a, double b) {
  dfx_d2u v; v.x = a; v.y = b;
  *reinterpret_cast<dfx_d2u*>(reinterpret_cast<char*>(base) + byte_off) = v;
}
// per-DOF array with 3 doubles per block, two lanes per block: lane 0 holds (x, y), lane 1 theta.  Lane 1 loads (y, theta): always in bounds.
__device__ __forceinline__ void ld3(const void* base, u32 o3, int e, double& a, double& b) {
  const double2 v = ldu2(base, o3);
  a = e ? v.y : v.x; b = v.y;
}
__device__ __forceinline__ void st3(void* base, u32 gb, int e, double a, double b) {
  if (e) stg<double>(base, gb * 24 + 16, a); else stu2(base, gb * 24, a, b);
}

// ---- who a lane is ----------------------------------------------------------------------------------------------------------------
template <int NW>
struct TileLane {
  static constexpr int TH = 2 * NW - 1, NBL = kTileW * TH, LW = kTileW + 2, NRING = 2 * LW + 2 * TH;
  int row0, col0, e;
  int hr, hc, lig_e;      // home block (tile-local; -1 / TH / 16: ring) and which of its ligaments this lane evaluates
  bool blk_lane, home_ok;
  int gb;                 // global block index of the home block (0 when there is none)
  __device__ __forceinline__ TileLane(const LigCtx& lc, int lwg) {
    const int ty = lwg / lc.tiles_x, tx = lwg - ty * lc.tiles_x;
    row0 = ty * TH; col0 = tx * kTileW;
    const int tid = threadIdx.x, P = tid >> 1;
    e = tid & 1; lig_e = e;
    blk_lane = P < NBL;
    bool role = true;
    if (blk_lane) { hr = P >> 4; hc = P & 15; }
    else {
      const int q = tid - 2 * NBL;
      if (q < TH) { hr = q; hc = -1; lig_e = 0; }                                            // enters the tile's left column
      else if (q < TH + kTileW) { hr = -1; hc = q - TH - lc.dc1; lig_e = 1; }                // enters its bottom row
      else if (lc.dc1 != 0 && q < 2 * TH + kTileW - 1) { hr = q - (TH + kTileW); hc = lc.dc1 < 0 ? kTileW : -1; lig_e = 1; }   // diagonal: from the side
      else { hr = 0; hc = 0; role = false; }
    }
    const int grow = row0 + hr, gcol = col0 + hc;
    home_ok = role && grow >= 0 && grow < lc.n_rows && gcol >= 0 && gcol < lc.R;
    gb = home_ok ? grow * lc.R + gcol : 0;
  }
  __device__ __forceinline__ static int cell(int r, int c) { return (r + 1) * LW + (c + 1); }
  // ring entry h -> tile-local coordinates
  __device__ __forceinline__ static void ring(int h, int& r, int& c) {
    if (h < LW) { r = -1; c = h - 1; }
    else if (h < 2 * LW) { r = TH; c = h - LW - 1; }
    else if (h < 2 * LW + TH) { r = h - 2 * LW; c = -1; }
    else { r = h - 2 * LW - TH; c = kTileW; }
  }
  // lanes whose partner-end contributions enter this block: from the left (e = 0 of the block before) and from below (e = 1)
  __device__ __forceinline__ int src_left() const { return hc > 0 ? (int)threadIdx.x - e - 2 : 2 * NBL + hr; }
  __device__ __forceinline__ int src_below(int dc1) const {
    const int cs = hc - dc1;
    if (hr == 0) return 2 * NBL + TH + hc;
    if (cs >= 0 && cs < kTileW) return 2 * ((hr - 1) * kTileW + cs) + 1;
    return 2 * NBL + TH + kTileW + (hr - 1);
  }
};

struct LigIn {
  BlockRec<double> o, p;
  double lx, ly, l0, il0, ks, ksh, kr, phi1, phi2, am, ac, kc, sgn;
};
// what a ligament lane needs beyond the two node vectors, once the LDS copies are there
template <int CONTACT>
__device__ __forceinline__ void lig_resolve(const DevCtx& c, const LigCtx& lc, const MemberBases& B, int m, int lig, int tab, const double2 (*s_rec)[2],
                                            const double2 (*s_dict)[2], int co, int cp, LigIn& L) {
  const double2 o0 = s_rec[co][0], o1 = s_rec[co][1], p0 = s_rec[cp][0], p1 = s_rec[cp][1];
  L.o.x = o0.x; L.o.y = o0.y; L.o.th = o1.x; L.o.sh = o1.y; L.o.ch = half_cos(o1.x, o1.y);
  L.p.x = p0.x; L.p.y = p0.y; L.p.th = p1.x; L.p.sh = p1.y; L.p.ch = half_cos(p1.x, p1.y);
  const u32 n2 = (u32)c.n_blocks * 2;
  if (c.l_dict_lds) {
    const int li = (tab >> 8) & 0xff;
    const double2 lv = s_dict[li][0], ln = s_dict[li][1];
    L.lx = lv.x; L.ly = lv.y; L.l0 = ln.x; L.il0 = ln.y;
  } else {
    const double2 lv = ldg<double2>(lc.l + (size_t)m * n2 * 2, (u32)lig * 16);
    L.lx = lv.x; L.ly = lv.y; L.l0 = sqrt(lv.x * lv.x + lv.y * lv.y); L.il0 = 1.0 / L.l0;
  }
  const double* cst = B.cst;
  if (c.k_uniform) { L.ks = cst[3]; L.ksh = cst[4]; L.kr = cst[5]; }
  else {
    const double2 k01 = ldg<double2>(lc.k + (size_t)m * n2 * 4, (u32)lig * 32);
    L.ks = k01.x; L.ksh = k01.y; L.kr = ldg<double>(lc.k + (size_t)m * n2 * 4, (u32)lig * 32 + 16);
  }
  L.sgn = (tab & 2) ? 1.0 : -1.0;
  L.phi1 = L.phi2 = L.am = L.ac = L.kc = 0.0;
  if (CONTACT == 1) {
    L.am = cst[0]; L.ac = cst[1]; L.kc = cst[2];
    // culling bound of pack_params (cst[9], cst[10]), as in resolve_lane: the void angles are loaded only where the contact can engage
    double2 ph = make_double2(cst[10], cst[10]);
    if (!(fabs(L.o.th - L.p.th) <= cst[9])) ph = ldg<double2>(lc.phi + (size_t)m * n2 * 2, (u32)lig * 16);
    L.phi1 = ph.x; L.phi2 = ph.y;
  }
}

// ---- forward stage (arguments as k_fwd_stage; fixed grid, table build) --------------------------------------------------------------
#ifndef DFX_TILE_FWD_OCC
#define DFX_TILE_FWD_OCC
#endif
#ifndef DFX_TILE_ADJ_OCC
#define DFX_TILE_ADJ_OCC
#endif
template <int MODEL, int CONTACT, int NW>
__global__ __launch_bounds__(64 * NW) DFX_TILE_FWD_OCC void k_fwd_tile(DevCtx c, LigCtx lc, StageCoef sc, int i, int j, int in_buf, int out_buf, int y_buf, int mode) {
  typedef TileLane<NW> TL;
  __shared__ double2 s_rec[(TL::TH + 2) * TL::LW][2];
  __shared__ double s_con[64 * NW][3];
  __shared__ double2 s_dict[kDictLds][2];
  const int m = blockIdx.y + c.m0;
  const TL t(lc, logical_wg(blockIdx.x, lc.n_wg));
  const int tid = threadIdx.x, e = t.e;
  const u32 gb = (u32)t.gb, lig = 2 * gb + t.lig_e, n2 = (u32)c.n_blocks * 2, nd = (u32)c.n_blocks * 3;
  const bool blk = t.blk_lane && t.home_ok;
  const Seg sg = *c.cur;
  const long long n = sg.base_step + j;
  const int write_traj = mode & 1;
  // ---- load phase: everything is in flight before the first wait
  const double* POSin = pos_in(c, m, in_buf, n);
  const MemberBases B = member_bases(c, m);
  const int tab = ldg<int>(lc.tab + (size_t)m * n2, lig * 4);
  const double2 pa = ldg<double2>(lc.p + (size_t)m * n2 * 4, lig * 32), pb = ldg<double2>(lc.p + (size_t)m * n2 * 4, lig * 32 + 16);
  const double2 rec = ldg<double2>(POSin, (gb * kPos + 2 * e) * 8);
  if (c.l_dict_lds && tid < kDictLds) { s_dict[tid][0] = ldg<double2>(B.l_dict, (u32)tid * 32); s_dict[tid][1] = ldg<double2>(B.l_dict, (u32)tid * 32 + 16); }
  if (!t.blk_lane) {        // the crew: the ring of blocks around the tile
    const int q = tid - 2 * TL::NBL;
#pragma unroll
    for (int it = 0; it < (TL::NRING + 31) / 32; ++it) {
      int r, cc;
      TL::ring(q + 32 * it, r, cc);
      const int grow = t.row0 + r, gcol = t.col0 + cc;
      if (q + 32 * it < TL::NRING && grow >= 0 && grow < lc.n_rows && gcol >= 0 && gcol < lc.R) {
        const u32 o = (u32)(grow * lc.R + gcol) * (kPos * 8);
        const double2 a0 = ldg<double2>(POSin, o), a1 = ldg<double2>(POSin, o + 16);
        s_rec[TL::cell(r, cc)][0] = a0; s_rec[TL::cell(r, cc)][1] = a1;
      }
    }
  }
  const u32 o_rec = (gb * kPos + 2 * e) * 8, o3 = (gb * 3 + e) * 8;
  const double2 qn2 = ldg<double2>(pos_in(c, m, y_buf, n), o_rec);
  double vnA, vnB, viA, viB, imA, imB, dpA, dpB;
  ld3(vel_in(c, m, y_buf, n), o3, e, vnA, vnB);
  ld3(vel_in(c, m, in_buf, n), o3, e, viA, viB);
  ld3(c.inv_m + (size_t)m * nd, o3, e, imA, imB);
  if (c.damping_uniform) { dpA = e ? B.cst[8] : B.cst[6]; dpB = B.cst[7]; } else ld3(c.damping + (size_t)m * nd, o3, e, dpA, dpB);
  const int sidx = ldg<int>(c.block_special, gb * 4);
  const bool keep_stages = c.AD != nullptr;
  double* Am = keep_stages ? c.AD + (size_t)m * c.ad_stride + (size_t)n * ((u32)(c.s - 1) * nd) : c.A + (size_t)m * (u32)(c.s + 1) * nd;
  double svA = 0.0, sqA = 0.0, svB = 0.0, sqB = 0.0;
  {
    double2 al[kMaxStages - 1];
#pragma unroll
    for (int l = 0; l < kMaxStages - 1; ++l) al[l] = l < i ? ldu2(Am + (size_t)l * nd, o3) : make_double2(0.0, 0.0);
#pragma unroll
    for (int l = 0; l < kMaxStages - 1; ++l) {
      const double a = e ? al[l].y : al[l].x;
      svA += sc.cv[l] * a; sqA += sc.cq[l] * a;
      svB += sc.cv[l] * al[l].y; sqB += sc.cq[l] * al[l].y;
    }
  }
  if (blk) s_rec[TL::cell(t.hr, t.hc)][e] = rec;
  __syncthreads();
  // ---- the lane's ligament, once: own end into registers, the partner's end into LDS
  double fx = 0.0, fy = 0.0, fth = 0.0, pfx = 0.0, pfy = 0.0, pfth = 0.0;
  if (t.home_ok && (tab & 1)) {
    LigIn L;
    lig_resolve<CONTACT>(c, lc, B, m, (int)lig, tab, s_rec, s_dict, TL::cell(t.hr, t.hc), TL::cell(t.hr + t.lig_e, t.hc + (t.lig_e ? lc.dc1 : 1)), L);
    BondGrad<double> g;
    BondPartner<double> pg;
    bond_grad<MODEL, double>(L.o, L.p, pa.x, pa.y, pb.x, pb.y, L.lx, L.ly, L.l0, L.il0, L.ks, L.ksh, L.kr, L.sgn, g, &pg);
    fx = g.fx; fy = g.fy; fth = g.fth;
    pfx = -g.fx; pfy = -g.fy; pfth = pg.fth;
    if (CONTACT == 1) {
      ContactGrad<double> cg;
      contact_grad<double>(L.sgn * (L.o.th - L.p.th), L.phi1, L.phi2, L.am, L.ac, L.kc, cg);
      fth += L.sgn * cg.dkap;
      pfth -= L.sgn * cg.dkap;
    }
  }
  s_con[tid][0] = pfx; s_con[tid][1] = pfy; s_con[tid][2] = pfth;
  __syncthreads();
  // ---- block sums: own e = 0, own e = 1, from the left, from below (fixed order)
  const double rA = dpp_mov<0xB1>(e ? fx : fth), rB = dpp_mov<0xB1>(e ? fy : 0.0);
  if (!blk) return;
  const double* cl = s_con[t.src_left()];
  const double* cb = s_con[t.src_below(lc.dc1)];
  const double dEA = e ? ((rA + fth) + cl[2]) + cb[2] : ((fx + rA) + cl[0]) + cb[0];
  const double dEB = ((fy + rB) + cl[1]) + cb[1];
  // ---- DOF epilogue: lane 0 of the pair (x, y), lane 1 theta
  double h = sg.h;
  if (c.t_steps) { const double* ts = steps_of(c, m); h = ts[n + 1] - ts[n]; }
  const bool has_out = out_buf != -1;
  double aA, aB = 0.0, qA, qB = 0.0, vA, vB = 0.0;
  auto dof = [&](int k, double dE, double qn, double vn, double v_i, double invm, double damp, double sv, double sq, double& a, double& qnext, double& vnext) {
    bool constrained = false;
    double fload = 0.0;
    if (sidx >= 0) {
      const dfx_special& sp = c.special[sidx];
      constrained = (sp.con_mask >> k) & 1;
      if (!constrained && c.n_fns) {
        const double* ft_i = fn_tab_row(c, m, j, i);
        const u32 z = lane_zero();
        for (int f = 0; f < c.n_fns; ++f) fload += sp.load_coef[k][f] * fn_tab_get(ft_i, f, 0, z);
      }
    }
    a = constrained ? 0.0 : (fload - dE - damp * v_i) * invm;
    sv += sc.cv[i] * a;
    sq += sc.cq[i] * a;
    qnext = qn + h * (sc.c_next * vn + h * sq);
    vnext = vn + h * sv;
    if (constrained && has_out) {
      const dfx_special& sp = c.special[sidx];
      const double* ft_n = fn_tab_row(c, m, j, i + 1);
      const u32 z = lane_zero();
      qnext = 0.0; vnext = 0.0;
      for (int f = 0; f < c.n_fns; ++f) { qnext += sp.con_coef[k][f] * fn_tab_get(ft_n, f, 0, z); vnext += sp.con_coef[k][f] * fn_tab_get(ft_n, f, 1, z); }
    }
  };
  dof(e ? 2 : 0, dEA, qn2.x, vnA, viA, imA, dpA, svA, sqA, aA, qA, vA);
  if (!e) dof(1, dEB, qn2.y, vnB, viB, imB, dpB, svB, sqB, aB, qB, vB);
  if (!(keep_stages && i == c.s - 1)) st3(Am + (size_t)i * nd, gb, e, aA, aB);
  if (!has_out) return;
  // ---- publish the next stage record: each lane of the pair stores its own aligned 16-byte chunk (x, y) / (th, sin th/2)
  double2 chunk = make_double2(qA, qB);
  if (e) { double sn, cs; fast_sincos(0.5 * qA, &sn, &cs); chunk.y = sn; }
  if (out_buf >= 0) {
    stg<double2>(c.POS + ((size_t)m * c.nbuf + out_buf) * (u32)c.n_blocks * kPos, o_rec, chunk);
    st3(c.VEL + ((size_t)m * c.nbuf + out_buf) * nd, gb, e, vA, vB);
  }
  if (write_traj || out_buf < -1) {
    double* tr = out_buf < -1 ? traj_rec(c, m, out_buf, n) : traj_rec(c, m, -1, n + 1);
    stg_s<double2>(tr, o_rec, chunk);
    st3(tr + (size_t)c.n_blocks * kPos, gb, e, vA, vB);
  }
}

// ---- reverse stage (records build of k_adj_stage: no rebuild, no per-ligament gradients; arguments as there) --------------------------
template <int MODEL, int CONTACT, int NW>
__global__ __launch_bounds__(64 * NW) DFX_TILE_ADJ_OCC void k_adj_tile(DevCtx c, LigCtx lc, AdjCoef ac, int i, int j, int in_buf) {
  typedef TileLane<NW> TL;
  __shared__ double2 s_rec[(TL::TH + 2) * TL::LW][2];
  __shared__ double s_w[(TL::TH + 2) * TL::LW][3];
  __shared__ double s_con[64 * NW][6];
  __shared__ double2 s_dict[kDictLds][2];
  const int m = blockIdx.y + c.m0;
  const TL t(lc, logical_wg(blockIdx.x, lc.n_wg));
  const int tid = threadIdx.x, e = t.e;
  const u32 gb = (u32)t.gb, lig = 2 * gb + t.lig_e, n2 = (u32)c.n_blocks * 2, nd = (u32)c.n_blocks * 3, nd6 = (u32)c.n_blocks * 6;
  const bool blk = t.blk_lane && t.home_ok;
  const Seg sg = *c.cur;
  const long long n = sg.base_step + j;
  const int win = (int)((n * c.s + i) & 1);
  double h = sg.h, h_before = (sg.j0 + j) == 0 ? sg.h_prev : sg.h;
  if (c.t_steps) { const double* ts = steps_of(c, m); h = ts[n + 1] - ts[n]; h_before = n > 0 ? ts[n] - ts[n - 1] : 0.0; }
  // ---- load phase
  const double* POSin = pos_in(c, m, in_buf, n);
  const MemberBases B = member_bases(c, m);
  const double* Win = c.W + ((size_t)m * 2 + win) * nd;
  const int tab = ldg<int>(lc.tab + (size_t)m * n2, lig * 4);
  const double2 pa = ldg<double2>(lc.p + (size_t)m * n2 * 4, lig * 32), pb = ldg<double2>(lc.p + (size_t)m * n2 * 4, lig * 32 + 16);
  const double2 rec = ldg<double2>(POSin, (gb * kPos + 2 * e) * 8);
  if (c.l_dict_lds && tid < kDictLds) { s_dict[tid][0] = ldg<double2>(B.l_dict, (u32)tid * 32); s_dict[tid][1] = ldg<double2>(B.l_dict, (u32)tid * 32 + 16); }
  if (!t.blk_lane) {        // the crew: records and w (the copy the previous launch stored) of the ring
    const int q = tid - 2 * TL::NBL;
#pragma unroll
    for (int it = 0; it < (TL::NRING + 31) / 32; ++it) {
      int r, cc;
      TL::ring(q + 32 * it, r, cc);
      const int grow = t.row0 + r, gcol = t.col0 + cc;
      if (q + 32 * it < TL::NRING && grow >= 0 && grow < lc.n_rows && gcol >= 0 && gcol < lc.R) {
        const u32 bb = (u32)(grow * lc.R + gcol);
        const double2 a0 = ldg<double2>(POSin, bb * (kPos * 8)), a1 = ldg<double2>(POSin, bb * (kPos * 8) + 16);
        const double2 wxy = ldu2(Win, bb * 24);
        const double wth = ldg<double>(Win, bb * 24 + 16);
        const int cl_ = TL::cell(r, cc);
        s_rec[cl_][0] = a0; s_rec[cl_][1] = a1;
        s_w[cl_][0] = wxy.x; s_w[cl_][1] = wxy.y; s_w[cl_][2] = wth;
      }
    }
  }
  double* gm = lc.g + (size_t)m * n2 * 4;
  const double2 g_old0 = ldg_s<double2>(gm, lig * 32), g_old1 = ldg_s<double2>(gm, lig * 32 + 16);
  const u32 o3 = (gb * 3 + e) * 8, o6A = (gb * 6 + (e ? 4 : 0)) * 8, o6B = (gb * 6 + 2) * 8;
  double viA, viB, imA, imB, dpA, dpB, bmA, bmB, bcA = 0.0, bcB = 0.0;
  ld3(vel_in(c, m, in_buf, n), o3, e, viA, viB);
  ld3(c.inv_m + (size_t)m * nd, o3, e, imA, imB);
  if (c.damping_uniform) { dpA = e ? B.cst[8] : B.cst[6]; dpB = B.cst[7]; } else ld3(c.damping + (size_t)m * nd, o3, e, dpA, dpB);
  double* bmm = c.blk_m + (size_t)m * nd;
  double* bcm = c.blk_c + (size_t)m * nd;
  ld3(bmm, o3, e, bmA, bmB);
  if (c.blk_c) ld3(bcm, o3, e, bcA, bcB);
  const int sidx = ldg<int>(c.block_special, gb * 4);
  double* YBm = c.YB + (size_t)m * (u32)c.s * nd6;
  double* LAMm = c.LAM + (size_t)m * nd6;
  double lqA = 0.0, lvA = 0.0, lqB = 0.0, lvB = 0.0;
  if (i == 0 || ac.col[c.s] != 0.0 || ac.cur[c.s] != 0.0) {
    const double2 a = ldg_s<double2>(LAMm, o6A);
    lqA = a.x; lvA = a.y;
    if (!e) { const double2 b = ldg_s<double2>(LAMm, o6B); lqB = b.x; lvB = b.y; }
  }
  double sqA = 0.0, svA = 0.0, sqcA = 0.0, svcA = 0.0, sqB = 0.0, svB = 0.0, sqcB = 0.0, svcB = 0.0;
  {
    double2 ya[kMaxStages], yb[kMaxStages];
#pragma unroll
    for (int jj = 1; jj < kMaxStages; ++jj) {
      const bool on = jj > i && jj < c.s;
      ya[jj] = on ? ldg_s<double2>(YBm + (size_t)jj * nd6, o6A) : make_double2(0.0, 0.0);
      yb[jj] = (on && !e) ? ldg_s<double2>(YBm + (size_t)jj * nd6, o6B) : make_double2(0.0, 0.0);
    }
#pragma unroll
    for (int jj = 1; jj < kMaxStages; ++jj) {
      const double cf = i > 0 ? ac.col[jj] : 1.0;
      sqA += cf * ya[jj].x; svA += cf * ya[jj].y; sqcA += ac.cur[jj] * ya[jj].x; svcA += ac.cur[jj] * ya[jj].y;
      sqB += cf * yb[jj].x; svB += cf * yb[jj].y; sqcB += ac.cur[jj] * yb[jj].x; svcB += ac.cur[jj] * yb[jj].y;
    }
  }
  // own w = Kbar_v / m of this stage, recomputed (as in the records build of k_adj_stage); the ring has the stored copy
  const double wA = (h * (ac.cur[c.s] * lvA + svcA)) * imA, wB = (h * (ac.cur[c.s] * lvB + svcB)) * imB;
  if (blk) {
    const int cl_ = TL::cell(t.hr, t.hc);
    s_rec[cl_][e] = rec;
    s_w[cl_][e ? 2 : 0] = wA;
    if (!e) s_w[cl_][1] = wB;
  }
  __syncthreads();
  // ---- the lane's ligament, once, in dual numbers: Hessian-vector product and parameter derivatives of both ends
  double hx = 0.0, hy = 0.0, hth = 0.0, ex = 0.0, ey = 0.0, eth = 0.0;
  double phx = 0.0, phy = 0.0, phth = 0.0, pex = 0.0, pey = 0.0, peth = 0.0;
  if (t.home_ok && (tab & 1)) {
    LigIn L;
    const int co = TL::cell(t.hr, t.hc), cp = TL::cell(t.hr + t.lig_e, t.hc + (t.lig_e ? lc.dc1 : 1));
    lig_resolve<CONTACT>(c, lc, B, m, (int)lig, tab, s_rec, s_dict, co, cp, L);
    const BlockRec<Dual> o = seed_rec(L.o, s_w[co][0], s_w[co][1], s_w[co][2]);
    const BlockRec<Dual> p = seed_rec(L.p, s_w[cp][0], s_w[cp][1], s_w[cp][2]);
    BondGrad<Dual> g;
    BondPartner<Dual> pg;
    bond_grad<MODEL, Dual>(o, p, pa.x, pa.y, pb.x, pb.y, L.lx, L.ly, L.l0, L.il0, L.ks, L.ksh, L.kr, L.sgn, g, &pg);
    hx = g.fx.e; hy = g.fy.e; hth = g.fth.e; ex = g.fx.v; ey = g.fy.v; eth = g.fth.v;
    phx = -g.fx.e; phy = -g.fy.e; phth = pg.fth.e; pex = -g.fx.v; pey = -g.fy.v; peth = pg.fth.v;
    double d_p1 = 0.0, d_p2 = 0.0;
    if (CONTACT == 1) {
      ContactGrad<Dual> cg;
      contact_grad<Dual>(L.sgn * (o.th - p.th), L.phi1, L.phi2, L.am, L.ac, L.kc, cg);
      hth += L.sgn * cg.dkap.e; eth += L.sgn * cg.dkap.v;
      phth -= L.sgn * cg.dkap.e; peth -= L.sgn * cg.dkap.v;
      d_p1 = cg.p1.e; d_p2 = cg.p2.e;
    }
    if (t.blk_lane) {      // the owner's tile accumulates (a ligament that enters from outside is accumulated by the tile of its owner)
      // L += w . F = -w . grad E   =>   dL/dp = -eps(dE/dp)
      stg_s<double2>(gm, lig * 32, make_double2(g_old0.x - g.rx.e, g_old0.y - g.ry.e));
      stg_s<double2>(gm, lig * 32 + 16, make_double2(g_old1.x - pg.rx.e, g_old1.y - pg.ry.e));
      if (CONTACT == 1 && (d_p1 != 0.0 || d_p2 != 0.0)) {     // contacts are rare: the void-angle accumulator moves only where one is engaged
        double* gp = lc.gphi + (size_t)m * n2 * 2;
        const double2 po = ldg<double2>(gp, lig * 16);
        stg<double2>(gp, lig * 16, make_double2(po.x - d_p1, po.y - d_p2));
        c.touch[0] = 1;
      }
    }
  }
  { double* sc_ = s_con[tid]; sc_[0] = phx; sc_[1] = phy; sc_[2] = phth; sc_[3] = pex; sc_[4] = pey; sc_[5] = peth; }
  __syncthreads();
  // ---- block sums (fixed order: own e = 0, own e = 1, from the left, from below)
  const double r0 = dpp_mov<0xB1>(e ? hx : hth), r1 = dpp_mov<0xB1>(e ? hy : 0.0), r2 = dpp_mov<0xB1>(e ? ex : eth), r3 = dpp_mov<0xB1>(e ? ey : 0.0);
  if (!blk) return;
  const double* cl = s_con[t.src_left()];
  const double* cb = s_con[t.src_below(lc.dc1)];
  const double hwA = e ? ((r0 + hth) + cl[2]) + cb[2] : ((hx + r0) + cl[0]) + cb[0];
  const double hwB = ((hy + r1) + cl[1]) + cb[1];
  const double dEA = e ? ((r2 + eth) + cl[5]) + cb[5] : ((ex + r2) + cl[3]) + cb[3];
  const double dEB = ((ey + r3) + cl[4]) + cb[4];
  // ---- DOF epilogue
  struct Out { double ybq, ybv, lq, lv, wout, bm, bc; };
  auto dof = [&](int k, double hw, double dE, double w_d, double v_i, double invm, double damp, double lq, double lv, double sq, double sv, double sqc,
                 double bm_old, double bc_old, Out& o) {
    bool constrained = false;
    double fload = 0.0;
    if (sidx >= 0) {
      const dfx_special& sp = c.special[sidx];
      constrained = (sp.con_mask >> k) & 1;
      const double* ft = fn_tab_row(c, m, j, i);
      double gp[kMaxFnParams];
      for (int f = 0; f < c.n_fns; ++f) {
        const double coef = constrained ? -hw * sp.con_coef[k][f] : w_d * sp.load_coef[k][f];
        const bool loaded = !constrained && sp.load_coef[k][f] != 0.0;
        if ((coef != 0.0 && c.fn_g) || loaded) {
          const u32 z = lane_zero();
          const double g = fn_tab_get(ft, f, 0, z);
          if (loaded) fload += sp.load_coef[k][f] * g;
          if (coef != 0.0 && c.fn_g) {
            for (int kk = 0; kk < kMaxFnParams; ++kk) gp[kk] = fn_tab_get(ft, f, 2 + kk, z);
            double* q = c.fn_g + (((size_t)m * c.n_special + sidx) * DFX_MAX_FNS + f) * DFX_FN_PARAMS;
            for (int kk = 0; kk < DFX_FN_PARAMS; ++kk) acc_add(q + kk, coef * gp[kk]);
          }
        }
      }
    }
    const double a_i = constrained ? 0.0 : (fload - dE - damp * v_i) * invm;
    const double kq_in = h * (ac.cur[c.s] * lq + sqc);      // Kbar_q of this stage, recomputed (zero on constrained DOFs: their lambda and Ybar are)
    o.ybq = 0.0; o.ybv = 0.0; o.bm = bm_old; o.bc = bc_old;
    if (!constrained) {
      o.ybq = -hw;
      o.ybv = kq_in - damp * w_d;
      o.bm = bm_old - w_d * a_i;
      o.bc = bc_old - w_d * v_i;
    }
    double kv;
    if (i > 0) kv = h * (ac.col[c.s] * lv + ac.col[i] * o.ybv + sv);
    else {
      lq += o.ybq + sq;
      lv += o.ybv + sv;
      if ((sg.j0 + j) == 0 && c.G && !constrained) {
        const double* G = c.G + ((size_t)sg.interval * c.batch + m) * (size_t)nd6;
        lq += G[gb * 6 + k]; lv += G[gb * 6 + 3 + k];
      }
      if (constrained) { lq = 0.0; lv = 0.0; }
      kv = h_before * ac.col[c.s] * lv;
    }
    o.lq = lq; o.lv = lv;
    o.wout = constrained ? 0.0 : kv * invm;
  };
  Out A, Bo;
  Bo.ybq = Bo.ybv = Bo.lq = Bo.lv = Bo.wout = Bo.bm = Bo.bc = 0.0;
  dof(e ? 2 : 0, hwA, dEA, wA, viA, imA, dpA, lqA, lvA, sqA, svA, sqcA, bmA, bcA, A);
  if (!e) dof(1, hwB, dEB, wB, viB, imB, dpB, lqB, lvB, sqB, svB, sqcB, bmB, bcB, Bo);
  if (e) stg_s<double>(bmm, gb * 24 + 16, A.bm); else stu2(bmm, gb * 24, A.bm, Bo.bm);
  if (c.blk_c) st3(bcm, gb, e, A.bc, Bo.bc);
  stg_s<double2>(YBm + (size_t)i * nd6, o6A, make_double2(A.ybq, A.ybv));
  if (!e) stg_s<double2>(YBm + (size_t)i * nd6, o6B, make_double2(Bo.ybq, Bo.ybv));
  if (i == 0) {
    stg<double2>(LAMm, o6A, make_double2(A.lq, A.lv));
    if (!e) stg<double2>(LAMm, o6B, make_double2(Bo.lq, Bo.lv));
  }
  st3(c.W + ((size_t)m * 2 + (win ^ 1)) * nd, gb, e, A.wout, Bo.wout);
}

// ---- slot-major <-> ligament-major (once per set_params / once per sweep) --------------------------------------------------------------
//   lig_slots: (own slot, partner slot) of ligament lane 2 b + e, or -1
__global__ __launch_bounds__(kThreads) void k_lig_pack(DevCtx c, const int32_t* lig_slots, int32_t* tab, double* p, double* l, double* k, double* phi) {
  const int m = blockIdx.y;
  const u32 lig = blockIdx.x * kThreads + threadIdx.x, n2 = (u32)c.n_blocks * 2;
  if (lig >= n2) return;
  const int so = lig_slots[2 * lig], sp = lig_slots[2 * lig + 1];
  const size_t o = (size_t)m * n2 + lig, ms = (size_t)m * (u32)c.n_slots;
  if (so < 0) { tab[o] = 0; for (int q = 0; q < 4; ++q) p[o * 4 + q] = 0.0; return; }
  const int info = c.slot_info[so];
  const int li = c.l_dict_on ? (int)c.p_lidx[ms + so] : 0;
  tab[o] = 1 | ((info & 1) << 1) | (li << 8);
  p[o * 4] = c.p_r[(ms + so) * 2]; p[o * 4 + 1] = c.p_r[(ms + so) * 2 + 1];
  p[o * 4 + 2] = c.p_r[(ms + sp) * 2]; p[o * 4 + 3] = c.p_r[(ms + sp) * 2 + 1];
  if (l) {
    if (c.l_dict_on) { l[o * 2] = c.l_dict[(size_t)m * 1024 + li * 4]; l[o * 2 + 1] = c.l_dict[(size_t)m * 1024 + li * 4 + 1]; }
    else { l[o * 2] = c.p_l[(ms + so) * 2]; l[o * 2 + 1] = c.p_l[(ms + so) * 2 + 1]; }
  }
  if (k) for (int q = 0; q < 4; ++q) k[o * 4 + q] = c.p_k[(ms + so) * 4 + q];
  if (phi) { phi[o * 2] = c.p_phi[(ms + so) * 2]; phi[o * 2 + 1] = c.p_phi[(ms + so) * 2 + 1]; }
}
// adds the ligament-major accumulators of a sweep to the slot-major ones the rest of the engine reads (every slot has one ligament)
__global__ __launch_bounds__(kThreads) void k_lig_unpack(DevCtx c, const int32_t* lig_slots, const double* g, const double* gphi) {
  const int m = blockIdx.y;
  const u32 lig = blockIdx.x * kThreads + threadIdx.x, n2 = (u32)c.n_blocks * 2;
  if (lig >= n2) return;
  const int so = lig_slots[2 * lig], sp = lig_slots[2 * lig + 1];
  if (so < 0) return;
  const size_t o = (size_t)m * n2 + lig, ms = (size_t)m * (u32)c.n_slots;
  c.g_r[(ms + so) * 2] += g[o * 4]; c.g_r[(ms + so) * 2 + 1] += g[o * 4 + 1];
  c.g_r[(ms + sp) * 2] += g[o * 4 + 2]; c.g_r[(ms + sp) * 2 + 1] += g[o * 4 + 3];
  if (gphi) {       // phi1 lives on the end-0 slot of a ligament, phi2 on its end-1 slot (k_pack_grads)
    const int end = c.slot_info[so] & 1;
    c.g_phi[ms + so] += end ? gphi[o * 2 + 1] : gphi[o * 2];
    c.g_phi[ms + sp] += end ? gphi[o * 2] : gphi[o * 2 + 1];
  }
}

}  // namespace
