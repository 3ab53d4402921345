// dfx_tile.h -- the two stage kernels with every ligament evaluated ONCE, on lattice tiles owned by ONE wavefront (round 4).
// Included by dfx_engine.hip after dfx_kernels.h.
//
// Why.  k_fwd_stage / k_adj_stage map one lane to one (block, node slot): every ligament is evaluated by both of its end lanes, each
// lane gathers its partner's record (and w) from memory, and every fourth lane idles in the DOF epilogue.  The counters say the stage
// launches keep the vector ALUs busy 65 - 75 % of a wave's lifetime in steady state, so instructions per block matter.  The reference
// evaluates a bond once (jax_md.smap.bond over the bond list, energy.py:179-197); so do these kernels:
//
//   * a wavefront owns a tile of 7 x 3 = 21 blocks of the lattice grid find_tiling recovers (block = row * R + col), THREE lanes per
//     block: lane k of a block is the DOF lane of component k (x, y, theta) in the integrator epilogue -- no idle fourth lane -- and
//     lanes k = 0, 1 evaluate the ligaments the block OWNS (k = 0: partner b + 1; k = 1: partner b + R + dc1, dc1 one of -1, 0, +1 per
//     lattice: quads 0, kagome -1).  Ownership = the end on the lower block id;
//   * the lanes k = 2 evaluate the ligaments that ENTER the tile from outside (owned by a block of the tile to the left / below: 10 of
//     them for quads) -- the only ligaments evaluated twice (10 of 52 per tile) -- without touching the accumulators;
//   * records (and, reverse, w = Kbar_v / m) of the tile and of the ring of 24 blocks around it live in a wave-private piece of LDS:
//     two extra load instructions per wave fetch the ring, no per-lane gathers from memory.  No workgroup barrier anywhere: a wave's
//     LDS operations complete in order, so its own stores are visible to its loads (the first version of this file used workgroup
//     tiles of 16 x 7 blocks with two barriers and was 17 - 22 % slower than the slot kernels: profiles/r04_tile_kernels.txt);
//   * one evaluation yields both ends' derivatives (bond_grad with BondPartner); both halves go through LDS to the DOF lanes of
//     the two blocks, summed in a fixed order (own k = 0, own k = 1, from the left, from below): bit-reproducible;
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

constexpr int kTW = 7, kTH = 3, kTB = kTW * kTH;          // tile: 7 x 3 blocks = 63 lanes
constexpr int kLW = kTW + 2, kCells = (kTH + 2) * kLW;    // tile + ring: 9 x 5 cells
constexpr int kRing = kCells - kTB;                       // 24 ring blocks
constexpr int kTileWaves = 2;                             // independent wavefronts per workgroup

struct LigCtx {
  int R, n_rows, dc1, tiles_x;      // grid of the lattice: block = row * R + col; column offset of the k = 1 partner; tiles per lattice row
  int n_tiles, n_wg;                // tiles / workgroups per member
  unsigned inv_tiles_x, pad0;       // ceil(2^32 / tiles_x): tile / tiles_x = umulhi(tile, inv_tiles_x)
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

// ---- who a lane is ----------------------------------------------------------------------------------------------------------------
struct WaveTile {
  int ln, wv, blk, k, r, c;   // lane, wave of the workgroup, block of the tile, lane of the block (= DOF), tile-local row / column
  int row0, col0;
  bool tile_ok, dof_ok;       // dof_ok: this lane is DOF lane k of a block of the lattice
  u32 gb;                     // that block
  int hr, hc, lig_e;          // the ligament this lane evaluates: its owner block (tile-local; -1 / kTH / kTW: ring) and direction
  bool lig_ok;                // the owner is a block of the lattice (whether it has a ligament there: LigCtx::tab)
  u32 go;                     // the owner block
  __device__ __forceinline__ WaveTile(const LigCtx& lc, int lwg) {
    ln = threadIdx.x & 63; wv = threadIdx.x >> 6;
    const int tile = lwg * kTileWaves + wv;
    tile_ok = tile < lc.n_tiles;
    const int ty = (int)__umulhi((unsigned)tile, lc.inv_tiles_x), tx = tile - ty * lc.tiles_x;
    row0 = ty * kTH; col0 = tx * kTW;
    blk = (ln * 171) >> 9; k = ln - 3 * blk;           // lane 63: blk = 21 -> no role
    r = (blk * 37) >> 8; c = blk - kTW * r;
    const bool lane_ok = tile_ok && ln < 3 * kTB;
    dof_ok = lane_ok && row0 + r < lc.n_rows && col0 + c < lc.R;
    gb = dof_ok ? (u32)((row0 + r) * lc.R + col0 + c) : 0u;
    bool role = lane_ok;
    hr = r; hc = c; lig_e = k;
    if (k == 2) {             // ligaments that enter the tile: from the left, from below, (diagonal lattices) from the side
      if (blk < kTH) { hr = blk; hc = -1; lig_e = 0; }
      else if (blk < kTH + kTW) { hr = -1; hc = blk - kTH - lc.dc1; lig_e = 1; }
      else if (lc.dc1 != 0 && blk < 2 * kTH + kTW - 1) { hr = blk - (kTH + kTW); hc = lc.dc1 < 0 ? kTW : -1; lig_e = 1; }
      else role = false;
    }
    const int grow = row0 + hr, gcol = col0 + hc;
    lig_ok = role && grow >= 0 && grow < lc.n_rows && gcol >= 0 && gcol < lc.R;
    go = lig_ok ? (u32)(grow * lc.R + gcol) : 0u;
  }
  __device__ __forceinline__ static int cell(int r, int c) { return (r + 1) * kLW + (c + 1); }
  // ring cell h (0 .. kRing - 1) -> tile-local coordinates: bottom row, top row, left column, right column
  __device__ __forceinline__ static void ring(int h, int& r, int& c) {
    if (h < kLW) { r = -1; c = h - 1; }
    else if (h < 2 * kLW) { r = kTH; c = h - kLW - 1; }
    else if (h < 2 * kLW + kTH) { r = h - 2 * kLW; c = -1; }
    else { r = h - 2 * kLW - kTH; c = kTW; }
  }
  // lanes whose partner-end halves enter this block: from the left (k = 0 of the block before) and from below (k = 1)
  __device__ __forceinline__ int src_left() const { return c > 0 ? 3 * (blk - 1) : 3 * r + 2; }
  __device__ __forceinline__ int src_below(int dc1) const {
    const int cs = c - dc1;
    if (r == 0) return 3 * (kTH + c) + 2;
    if (cs >= 0 && cs < kTW) return 3 * ((r - 1) * kTW + cs) + 1;
    return 3 * (kTH + kTW + r - 1) + 2;
  }
};

struct LigIn {
  BlockRec<double> o, p;
  double lx, ly, l0, il0, ks, ksh, kr, phi1, phi2, am, ac, kc, sgn;
};
// what a ligament lane needs beyond the two node vectors, once the LDS copies are there
// A wave's LDS operations complete in order, but the COMPILER must be told that the wave-private exchange below is one: a release /
// acquire pair at wavefront scope around a wave barrier (no instruction: s_barrier stays out of these kernels) keeps it from moving
// the loads of one phase above the stores of the previous one (round-4 advice: under the HIP memory model the unfenced form is a race).
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

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
template <int MODEL, int CONTACT>
__global__ __launch_bounds__(64 * kTileWaves) void k_fwd_tile(DevCtx c, LigCtx lc, StageCoef sc, int i, int j, int in_buf, int out_buf, int y_buf, int mode) {
  __shared__ double2 s_rec_[kTileWaves][kCells][2];
  __shared__ double s_con_[kTileWaves][64][2][3];           // [lane][own end | partner end][x y theta]
  __shared__ double2 s_dict_[kTileWaves][kDictLds][2];
  const int m = blockIdx.y + c.m0;
  const WaveTile t(lc, logical_wg(blockIdx.x, lc.n_wg));
  if (!t.tile_ok) return;                                    // (wave-uniform)
  double2 (*s_rec)[2] = s_rec_[t.wv];
  double (*s_con)[2][3] = s_con_[t.wv];
  double2 (*s_dict)[2] = s_dict_[t.wv];
  const int k = t.k;
  const u32 gb = t.gb, lig = 2 * t.go + t.lig_e, n2 = (u32)c.n_blocks * 2, nd = (u32)c.n_blocks * 3;
  const Seg sg = *c.cur;
  const long long n = sg.base_step + j;
  const int write_traj = mode & 1;
  // ---- load phase: everything is in flight before the first wait
  const double* POSin = pos_in(c, m, in_buf, n);
  const MemberBases B = member_bases(c, m);
  const int tab = ldg<int>(lc.tab + (size_t)m * n2, lig * 4);
  const double2 pa = ldg<double2>(lc.p + (size_t)m * n2 * 4, lig * 32), pb = ldg<double2>(lc.p + (size_t)m * n2 * 4, lig * 32 + 16);
  const u32 o_chunk = (gb * kPos + 2 * (k & 1)) * 8;        // lanes 0 / 1 of a block: its (x, y) / (theta, sin theta/2) chunk
  const double2 rec = ldg<double2>(POSin, o_chunk);
  if (t.ln < 2 * kRing) {                                   // the ring of blocks around the tile: one 16-byte chunk per lane
    int r, cc;
    WaveTile::ring(t.ln >> 1, r, cc);
    const int grow = t.row0 + r, gcol = t.col0 + cc;
    if (grow >= 0 && grow < lc.n_rows && gcol >= 0 && gcol < lc.R)
      s_rec[WaveTile::cell(r, cc)][t.ln & 1] = ldg<double2>(POSin, ((u32)(grow * lc.R + gcol) * kPos + 2 * (t.ln & 1)) * 8);
  }
  if (c.l_dict_lds && t.ln < kDictLds) { s_dict[t.ln][0] = ldg<double2>(B.l_dict, (u32)t.ln * 32); s_dict[t.ln][1] = ldg<double2>(B.l_dict, (u32)t.ln * 32 + 16); }
  const u32 o_dof = (gb * 3 + k) * 8, o_rec = (gb * kPos + k) * 8;
  const double qn = ldg<double>(pos_in(c, m, y_buf, n), o_rec);
  const double vn = ldg<double>(vel_in(c, m, y_buf, n), o_dof);
  const double v_i = ldg<double>(vel_in(c, m, in_buf, n), o_dof);
  const double invm = ldg<double>(c.inv_m + (size_t)m * nd, o_dof);
  const double damp = c.damping_uniform ? B.cst[6 + k] : ldg<double>(c.damping + (size_t)m * nd, o_dof);
  const int sidx = ldg<int>(c.block_special, gb * 4);
  const bool keep_stages = c.AD != nullptr;
  double* Am = keep_stages ? c.AD + (size_t)m * c.ad_stride + (size_t)n * ((u32)(c.s - 1) * nd) : c.A + (size_t)m * (u32)(c.s + 1) * nd;
  double al[kMaxStages - 1];
#pragma unroll
  for (int l = 0; l < kMaxStages - 1; ++l) al[l] = l < i ? ldg<double>(Am + (size_t)l * nd, o_dof) : 0.0;
  double sv = 0.0, sq = 0.0;
#pragma unroll
  for (int l = 0; l < kMaxStages - 1; ++l) { sv += sc.cv[l] * al[l]; sq += sc.cq[l] * al[l]; }
  if (t.dof_ok && k < 2) s_rec[WaveTile::cell(t.r, t.c)][k] = rec;
  // ---- the lane's ligament, once (a wave's LDS operations complete in order: the records above are visible without a barrier)
  double fx = 0.0, fy = 0.0, fth = 0.0, pfx = 0.0, pfy = 0.0, pfth = 0.0;
  if (t.lig_ok && (tab & 1)) {
    LigIn L;
    wave_lds_fence();
    lig_resolve<CONTACT>(c, lc, B, m, (int)lig, tab, s_rec, s_dict, WaveTile::cell(t.hr, t.hc), WaveTile::cell(t.hr + t.lig_e, t.hc + (t.lig_e ? lc.dc1 : 1)), L);
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
  { double (*q)[3] = s_con[t.ln]; q[0][0] = fx; q[0][1] = fy; q[0][2] = fth; q[1][0] = pfx; q[1][1] = pfy; q[1][2] = pfth; }
  wave_lds_fence();
  if (!t.dof_ok) return;
  // ---- block sum of component k: own ligament 0, own ligament 1, from the left, from below (fixed order)
  const double dE = ((s_con[3 * t.blk][0][k] + s_con[3 * t.blk + 1][0][k]) + s_con[t.src_left()][1][k]) + s_con[t.src_below(lc.dc1)][1][k];
  // ---- DOF epilogue (every lane of a block owns one DOF)
  double h = sg.h;
  if (c.t_steps) { const double* ts = steps_of(c, m); h = ts[n + 1] - ts[n]; }
  const bool has_out = out_buf != -1;
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
  const double a = constrained ? 0.0 : (fload - dE - damp * v_i) * invm;
  if (!(keep_stages && i == c.s - 1)) stg<double>(Am + (size_t)i * nd, o_dof, a);
  sv += sc.cv[i] * a;
  sq += sc.cq[i] * a;
  double qnext = qn + h * (sc.c_next * vn + h * sq), vnext = vn + h * sv;
  if (constrained && has_out) {
    const dfx_special& sp = c.special[sidx];
    const double* ft_n = fn_tab_row(c, m, j, i + 1);
    const u32 z = lane_zero();
    qnext = 0.0; vnext = 0.0;
    for (int f = 0; f < c.n_fns; ++f) { qnext += sp.con_coef[k][f] * fn_tab_get(ft_n, f, 0, z); vnext += sp.con_coef[k][f] * fn_tab_get(ft_n, f, 1, z); }
  }
  if (!has_out) return;
  // ---- publish the next stage record: (x, y) by lane 0, (theta, sin theta/2) by lane 2; the three velocities by their lanes.
  // Lane 0 needs y from lane 1 (the 3-lane groups straddle the 16-lane DPP rows: a wave-wide shuffle, not a row move)
  const double y1 = __shfl_down(qnext, 1, 64);
  double2 chunk = make_double2(qnext, y1);
  if (k == 2) { double sn, cs; fast_sincos(0.5 * qnext, &sn, &cs); chunk.y = sn; }
  const u32 o_out = (gb * kPos + (k == 2 ? 2 : 0)) * 8;
  if (out_buf >= 0) {
    if (k != 1) stg<double2>(c.POS + ((size_t)m * c.nbuf + out_buf) * (u32)c.n_blocks * kPos, o_out, chunk);
    stg<double>(c.VEL + ((size_t)m * c.nbuf + out_buf) * nd, o_dof, vnext);
  }
  if (write_traj || out_buf < -1) {
    double* tr = out_buf < -1 ? traj_rec(c, m, out_buf, n) : traj_rec(c, m, -1, n + 1);
    if (k != 1) stg_s<double2>(tr, o_out, chunk);
    stg_s<double>(tr + (size_t)c.n_blocks * kPos, o_dof, vnext);
  }
}

// ---- reverse stage (records build of k_adj_stage: no rebuild, no per-ligament gradients; arguments as there) --------------------------
template <int MODEL, int CONTACT>
__global__ __launch_bounds__(64 * kTileWaves) void k_adj_tile(DevCtx c, LigCtx lc, AdjCoef ac, int i, int j, int in_buf) {
  __shared__ double2 s_rec_[kTileWaves][kCells][2];
  __shared__ double s_w_[kTileWaves][kCells][3];
  __shared__ double2 s_con_[kTileWaves][64][2][3];          // [lane][own end | partner end][x y theta] = (H w, dE/du)
  __shared__ double2 s_dict_[kTileWaves][kDictLds][2];
  const int m = blockIdx.y + c.m0;
  const WaveTile t(lc, logical_wg(blockIdx.x, lc.n_wg));
  if (!t.tile_ok) return;
  double2 (*s_rec)[2] = s_rec_[t.wv];
  double (*s_w)[3] = s_w_[t.wv];
  double2 (*s_con)[2][3] = s_con_[t.wv];
  double2 (*s_dict)[2] = s_dict_[t.wv];
  const int k = t.k;
  const u32 gb = t.gb, lig = 2 * t.go + t.lig_e, n2 = (u32)c.n_blocks * 2, nd = (u32)c.n_blocks * 3, nd6 = (u32)c.n_blocks * 6;
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
  const double2 rec = ldg<double2>(POSin, (gb * kPos + 2 * (k & 1)) * 8);
  if (t.ln < 2 * kRing) {                                   // the ring: records ...
    int r, cc;
    WaveTile::ring(t.ln >> 1, r, cc);
    const int grow = t.row0 + r, gcol = t.col0 + cc;
    if (grow >= 0 && grow < lc.n_rows && gcol >= 0 && gcol < lc.R)
      s_rec[WaveTile::cell(r, cc)][t.ln & 1] = ldg<double2>(POSin, ((u32)(grow * lc.R + gcol) * kPos + 2 * (t.ln & 1)) * 8);
  }
  if (t.ln < kRing) {                                       // ... and w (the copy the previous launch stored)
    int r, cc;
    WaveTile::ring(t.ln, r, cc);
    const int grow = t.row0 + r, gcol = t.col0 + cc;
    if (grow >= 0 && grow < lc.n_rows && gcol >= 0 && gcol < lc.R) {
      const u32 bb = (u32)(grow * lc.R + gcol) * 24;
      const double2 wxy = ldu2(Win, bb);
      const double wth = ldg<double>(Win, bb + 16);
      double* q = s_w[WaveTile::cell(r, cc)];
      q[0] = wxy.x; q[1] = wxy.y; q[2] = wth;
    }
  }
  if (c.l_dict_lds && t.ln < kDictLds) { s_dict[t.ln][0] = ldg<double2>(B.l_dict, (u32)t.ln * 32); s_dict[t.ln][1] = ldg<double2>(B.l_dict, (u32)t.ln * 32 + 16); }
  const u32 o_dof = (gb * 3 + k) * 8, o_b6 = (gb * 6 + 2 * k) * 8;
  const double v_i = ldg_s<double>(vel_in(c, m, in_buf, n), o_dof);
  const double invm = ldg<double>(c.inv_m + (size_t)m * nd, o_dof);
  const double damp = c.damping_uniform ? B.cst[6 + k] : ldg<double>(c.damping + (size_t)m * nd, o_dof);
  const int sidx = ldg<int>(c.block_special, gb * 4);
  double* YBm = c.YB + (size_t)m * (u32)c.s * nd6;
  double* LAMm = c.LAM + (size_t)m * nd6;
  double lq = 0.0, lv = 0.0, sq = 0.0, sv = 0.0, sqc = 0.0, svc = 0.0;
  if (i == 0 || ac.col[c.s] != 0.0 || ac.cur[c.s] != 0.0) { const double2 l2 = ldg_s<double2>(LAMm, o_b6); lq = l2.x; lv = l2.y; }
  {
    double2 yb[kMaxStages];
#pragma unroll
    for (int jj = 1; jj < kMaxStages; ++jj) yb[jj] = (jj > i && jj < c.s) ? ldg_s<double2>(YBm + (size_t)jj * nd6, o_b6) : make_double2(0.0, 0.0);
#pragma unroll
    for (int jj = 1; jj < kMaxStages; ++jj) {
      const double cf = i > 0 ? ac.col[jj] : 1.0;
      sq += cf * yb[jj].x; sv += cf * yb[jj].y; sqc += ac.cur[jj] * yb[jj].x; svc += ac.cur[jj] * yb[jj].y;
    }
  }
  // own w = Kbar_v / m of this stage, recomputed (as in the records build of k_adj_stage); the ring has the stored copy
  const double w_d = (h * (ac.cur[c.s] * lv + svc)) * invm;
  if (t.dof_ok) {
    const int cl_ = WaveTile::cell(t.r, t.c);
    if (k < 2) s_rec[cl_][k] = rec;
    s_w[cl_][k] = w_d;
  }
  // ---- the lane's ligament, once, in dual numbers: Hessian-vector product and parameter derivatives of both ends
  double2 own[3], par[3];
  for (int q = 0; q < 3; ++q) { own[q] = make_double2(0.0, 0.0); par[q] = make_double2(0.0, 0.0); }
  if (t.lig_ok && (tab & 1)) {
    LigIn L;
    const int co = WaveTile::cell(t.hr, t.hc), cp = WaveTile::cell(t.hr + t.lig_e, t.hc + (t.lig_e ? lc.dc1 : 1));
    wave_lds_fence();
    lig_resolve<CONTACT>(c, lc, B, m, (int)lig, tab, s_rec, s_dict, co, cp, L);
    const BlockRec<Dual> o = seed_rec(L.o, s_w[co][0], s_w[co][1], s_w[co][2]);
    const BlockRec<Dual> p = seed_rec(L.p, s_w[cp][0], s_w[cp][1], s_w[cp][2]);
    BondGrad<Dual> g;
    BondPartner<Dual> pg;
    bond_grad<MODEL, Dual>(o, p, pa.x, pa.y, pb.x, pb.y, L.lx, L.ly, L.l0, L.il0, L.ks, L.ksh, L.kr, L.sgn, g, &pg);
    own[0] = make_double2(g.fx.e, g.fx.v); own[1] = make_double2(g.fy.e, g.fy.v); own[2] = make_double2(g.fth.e, g.fth.v);
    par[0] = make_double2(-g.fx.e, -g.fx.v); par[1] = make_double2(-g.fy.e, -g.fy.v); par[2] = make_double2(pg.fth.e, pg.fth.v);
    double d_p1 = 0.0, d_p2 = 0.0;
    if (CONTACT == 1) {
      ContactGrad<Dual> cg;
      contact_grad<Dual>(L.sgn * (o.th - p.th), L.phi1, L.phi2, L.am, L.ac, L.kc, cg);
      own[2].x += L.sgn * cg.dkap.e; own[2].y += L.sgn * cg.dkap.v;
      par[2].x -= L.sgn * cg.dkap.e; par[2].y -= L.sgn * cg.dkap.v;
      d_p1 = cg.p1.e; d_p2 = cg.p2.e;
    }
    if (k < 2) {           // the owner's tile accumulates (a ligament that enters from outside is accumulated by the tile of its owner)
      // L += w . F = -w . grad E   =>   dL/dp = -eps(dE/dp).  Read-modify-write of the one lane that owns the ligament.
      double* gm = lc.g + (size_t)m * n2 * 4;
      const double2 g0 = ldg_s<double2>(gm, lig * 32), g1 = ldg_s<double2>(gm, lig * 32 + 16);
      stg_s<double2>(gm, lig * 32, make_double2(g0.x - g.rx.e, g0.y - g.ry.e));
      stg_s<double2>(gm, lig * 32 + 16, make_double2(g1.x - pg.rx.e, g1.y - pg.ry.e));
      if (CONTACT == 1 && (d_p1 != 0.0 || d_p2 != 0.0)) {     // contacts are rare: the void-angle accumulator moves only where one is engaged
        double* gp = lc.gphi + (size_t)m * n2 * 2;
        const double2 po = ldg<double2>(gp, lig * 16);
        stg<double2>(gp, lig * 16, make_double2(po.x - d_p1, po.y - d_p2));
        c.touch[0] = 1;
      }
    }
  }
  { double2 (*q)[3] = s_con[t.ln]; q[0][0] = own[0]; q[0][1] = own[1]; q[0][2] = own[2]; q[1][0] = par[0]; q[1][1] = par[1]; q[1][2] = par[2]; }
  wave_lds_fence();
  if (!t.dof_ok) return;
  // ---- block sums of component k (fixed order: own ligament 0, own ligament 1, from the left, from below)
  const double2 a0 = s_con[3 * t.blk][0][k], a1 = s_con[3 * t.blk + 1][0][k], a2 = s_con[t.src_left()][1][k], a3 = s_con[t.src_below(lc.dc1)][1][k];
  const double hw = ((a0.x + a1.x) + a2.x) + a3.x, dE = ((a0.y + a1.y) + a2.y) + a3.y;
  // ---- DOF epilogue (as k_adj_stage, records build)
  double* bmm = c.blk_m + (size_t)m * nd;
  double* bcm = c.blk_c + (size_t)m * nd;
  const double bm_old = ldg_s<double>(bmm, o_dof);
  const double bc_old = c.blk_c ? ldg<double>(bcm, o_dof) : 0.0;
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
        const double gv = fn_tab_get(ft, f, 0, z);
        if (loaded) fload += sp.load_coef[k][f] * gv;
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
  double ybq = 0.0, ybv = 0.0;
  if (!constrained) {
    ybq = -hw;
    ybv = kq_in - damp * w_d;
    stg_s<double>(bmm, o_dof, bm_old - w_d * a_i);
    if (c.blk_c) stg<double>(bcm, o_dof, bc_old - w_d * v_i);
  }
  stg_s<double2>(YBm + (size_t)i * nd6, o_b6, make_double2(ybq, ybv));
  double kv;
  if (i > 0) kv = h * (ac.col[c.s] * lv + ac.col[i] * ybv + sv);
  else {
    lq += ybq + sq;
    lv += ybv + sv;
    if ((sg.j0 + j) == 0 && c.G && !constrained) {
      const double* G = c.G + ((size_t)sg.interval * c.batch + m) * (size_t)nd6;
      lq += G[gb * 6 + k]; lv += G[gb * 6 + 3 + k];
    }
    if (constrained) { lq = 0.0; lv = 0.0; }
    stg<double2>(LAMm, o_b6, make_double2(lq, lv));
    kv = h_before * ac.col[c.s] * lv;
  }
  stg<double>(c.W + ((size_t)m * 2 + (win ^ 1)) * nd, o_dof, constrained ? 0.0 : kv * invm);
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
