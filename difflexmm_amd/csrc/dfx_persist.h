// dfx_persist.h -- the stage loop WITHOUT kernel boundaries, for solves whose launches do not fill the chip (round 5).
// Included by the engine after dfx_kernels.h (same anonymous namespace, same helpers, same per-ligament physics).
//
// Why.  One launch per Runge-Kutta stage is the right seam where a launch fills the chip (16 x 128x128: 15 / 29 us of work per launch
// against a 1.5 us boundary).  One 128x128 system is 1 024 waves -- one per SIMD -- and its stage launch is a 1.6 us dispatch plus ONE
// wave's dependent chain of memory round trips: 7.0 - 7.7 us per stage for 0.6 us of traffic (profiles/r04_v5_*: 0.09 - 0.18 of the
// roofline), the reference's own use (one design) and config 4's per-GPU width.  reference loop: odeint's while_loop x 6 rhs,
// /root/reference/difflexmm/dynamics.py:166.
//
// How.  A wave owns its 16 (20: packed triangles) blocks for a whole segment of <= 256 steps: parameters, step state and the stage
// accelerations stay in registers (reverse loop: lambda, the Ybar history and the gradient accumulators in lane-private LDS words, which
// is what lets three of its workgroups share a compute unit).  What the NEXT stage of a neighbouring wave needs is the 32-byte stage record of
// a block (x, y, theta, sin theta/2) -- nothing else crosses waves, and a wave has <= 4 neighbour waves on a lattice.  Hand-off (MI355X_MICROARCH.md,
// "Valid forms", R2: the data is the flag):
//   * records travel through a RING of kPRing places per member, [place][member][block][4 doubles]; the record of stage ordinal t of a
//     launch lives in place t % kPRing;
//   * every ring store is a 16-byte write-through store (sc1), every ring load a 16-byte sc1 load (served past the L1, which another
//     CU's stores never refresh);
//   * a place holds POISON (all-ones words: a NaN no arithmetic produces) until its record arrives; a lane polls its partner's record
//     until none of its four doubles is poison.  An aligned 8-byte word is written by one store and is never seen torn, so no ordering
//     between the stores, no flag and no fence is needed;
//   * the owner re-poisons the place of ordinal t + kPAhead while it works on stage t; the s_waitcnt of its next poll completes that
//     store long before a neighbour can ask for ordinal t + kPAhead (a neighbour reaches stage t only after this wave has finished
//     stage t - 2).  The places of ordinals 0 .. kPAhead-1 are poisoned by a small launch in front (k_ring_poison), ordinal 0 is
//     published by every wave at its start.  kPRing >= kPAhead + 1 keeps a re-poison from overtaking a reader.
//   * every spin is bounded; a wave that gives up stores a code the host turns into an error (workgroup not resident).
// The mock with this protocol (tools/mock/persistent_stage_mock.hip, profiles/r05_persistent_stage_mock.txt): 2.9 us per stage for one
// 128x128 system with 352 fp64 operations per lane, every word of every record equal to the one-launch-per-stage form, also under
// uneven load, 4 workgroups per compute unit, and with finite garbage in the unpoisoned places.
//
// Residency.  All waves of a launch must be resident at once: grid = ceil(waves / 4) workgroups of 256 threads (one wave per SIMD),
// which the dispatcher spreads evenly over the compute units by itself (measured with and without an LDS allocation that would force
// it: profiles/r05_persistent_stage_mock.txt run 5, profiles/r05_persistent_kernels.txt), and the host refuses shapes whose
// workgroups per compute unit exceed what the kernel's registers allow (persist_members_that_fit: members beyond it follow in a
// second launch of the same segment).  Persistent launches of different streams (several engines of one process) overlap only while
// their needs fit a compute unit together; otherwise the later one queues behind the other stream (engine_launch.hip, launch_persist):
// two half-resident launches would starve each other.  Checkpoint stores (records / state / stage accelerations) are plain stores
// beside the ring, in the layout the stage kernels write, so every reverse path reads them unchanged.
#pragma once
#include "dfx_persist_api.h"

namespace {
using namespace dfx_persist;

typedef unsigned p_u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bool is_poison(double x) { return (unsigned)__double2hiint(x) == kPoisonWord; }

// one lane's two 16-byte chunks of a ring record, sc1 loads, waited for inside the statement (the compiler does not count asm loads)
__device__ __forceinline__ void ring_load(const double* place, u32 byte_off, double (&r)[4]) {
  p_u4 c0, c1;
  asm volatile("global_load_dwordx4 %0, %2, %3 sc1\n\tglobal_load_dwordx4 %1, %2, %3 offset:16 sc1\n\ts_waitcnt vmcnt(0)"
               : "=&v"(c0), "=&v"(c1) : "v"(byte_off), "s"(place) : "memory");
  r[0] = __hiloint2double(c0.y, c0.x); r[1] = __hiloint2double(c0.w, c0.z);
  r[2] = __hiloint2double(c1.y, c1.x); r[3] = __hiloint2double(c1.w, c1.z);
}
__device__ __forceinline__ void ring_store(double* place, u32 byte_off, double a, double b) {
  p_u4 x;
  x.x = __double2loint(a); x.y = __double2hiint(a); x.z = __double2loint(b); x.w = __double2hiint(b);
  asm volatile("global_store_dwordx4 %0, %1, %2 sc1\n\ts_nop 1" ::"v"(byte_off), "v"(x), "s"(place) : "memory");
}
__device__ __forceinline__ void ring_poison(double* place, u32 byte_off) {
  p_u4 x; x.x = x.y = x.z = x.w = kPoisonWord;
  asm volatile("global_store_dwordx4 %0, %1, %2 sc1\n\ts_nop 1" ::"v"(byte_off), "v"(x), "s"(place) : "memory");
}

// places 0 .. kPAhead-1 of the members of a launch, poisoned in front of it (stream order)
__global__ __launch_bounds__(kThreads) void k_ring_poison(double* ring, int batch, int n_blocks, int m0, int width /* doubles per block */, int places) {
  const int m = blockIdx.y + m0;
  const int tid = blockIdx.x * kThreads + threadIdx.x;
  if (tid >= n_blocks * width) return;
  const unsigned long long p = ~0ull;
  for (int pl = 0; pl < places; ++pl)
    reinterpret_cast<unsigned long long*>(ring)[((size_t)pl * batch + m) * ((size_t)n_blocks * width) + tid] = p;
}

// what a lane keeps about its slot's ligament for the whole launch (the stage kernels load this at every stage: issue_lane / resolve_lane)
struct LigRes {
  double rox, roy, rpx, rpy, lx, ly, l0, il0, ks, ksh, kr, sgn;
  double am, ac, kc, phi1, phi2, kap_safe, phi_min;      // angle contact
  int info, pslot;
};
template <int CONTACT>
__device__ __forceinline__ void load_lig_res(const DevCtx& c, const MemberBases& B, int slot, LigRes& g) {
  g.info = ldg<int>(c.slot_info, (u32)slot * 4);
  g.pslot = g.info < 0 ? slot : (g.info >> 1);          // a slot without a ligament watches its own block
  g.sgn = (g.info & 1) ? 1.0 : -1.0;
  const double2 ro = ldg<double2>(B.p_r, (u32)slot * 16), rp = ldg<double2>(B.p_r, (u32)g.pslot * 16);
  g.rox = ro.x; g.roy = ro.y; g.rpx = rp.x; g.rpy = rp.y;
  if (c.l_dict_on) {
    const u32 li = (u32)ldg<uint8_t>(B.p_lidx, (u32)slot);
    const double2 lv = ldg<double2>(B.l_dict, li * 32), ln = ldg<double2>(B.l_dict, li * 32 + 16);
    g.lx = lv.x; g.ly = lv.y; g.l0 = ln.x; g.il0 = ln.y;
  } else {
    const double2 lv = ldg<double2>(B.p_l, (u32)slot * 16);
    g.lx = lv.x; g.ly = lv.y;
    g.l0 = g.info < 0 ? 1.0 : sqrt(lv.x * lv.x + lv.y * lv.y);
    g.il0 = 1.0 / g.l0;
  }
  if (c.k_uniform) { g.ks = B.cst[3]; g.ksh = B.cst[4]; g.kr = B.cst[5]; }
  else { g.ks = ldg<double>(B.p_k, (u32)slot * 32); g.ksh = ldg<double>(B.p_k, (u32)slot * 32 + 8); g.kr = ldg<double>(B.p_k, (u32)slot * 32 + 16); }
  g.am = g.ac = g.kc = g.phi1 = g.phi2 = g.kap_safe = g.phi_min = 0.0;
  if (CONTACT == 1) {
    g.am = B.cst[0]; g.ac = B.cst[1]; g.kc = B.cst[2]; g.kap_safe = B.cst[9]; g.phi_min = B.cst[10];
    const double2 ph = ldg<double2>(B.p_phi, (u32)slot * 16);
    g.phi1 = ph.x; g.phi2 = ph.y;
  }
}
// the partner's ring record of stage ordinal t: polled until none of its four doubles is poison; false: gave up
__device__ __forceinline__ bool ring_wait(const double* place, u32 byte_off, double (&r)[4], int t_ord, int* give_up, int limit) {
  for (int spins = 0;;) {
    ring_load(place, byte_off, r);
    const bool ok = !(is_poison(r[0]) || is_poison(r[1]) || is_poison(r[2]) || is_poison(r[3]));
    if (__all(ok)) return true;
    if (++spins > limit) { *give_up = 1 + t_ord; return false; }
    __builtin_amdgcn_s_sleep(1);
  }
}

#ifdef DFX_PERSIST_TIMING
// diagnostic build (make timing -> variants/libdfx_timing.so): where one wave's stage goes, in ticks of the shader clock counter, summed over a
// launch by wave 0 and left in the pinned words behind the give-up word (engine_forward.hip prints them)
__device__ __forceinline__ unsigned long long tick() {
  unsigned long long t;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
#if DFX_PERSIST_TIMING == 2      // light: the poll against everything else, no waits added
#define DFX_TICK(k) if ((k) < 4) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); acc_t[k] += (unsigned)(t_ - t_prev); t_prev = t_; }
#else
#define DFX_TICK(k) { const unsigned long long t_ = tick(); acc_t[k] += (unsigned)(t_ - t_prev); t_prev = t_; }
#endif
#else
#define DFX_TICK(k)
#endif

// which member and which of its waves this wave is.  Dense: waves packed one after the other.  XCD-aware (pa.xcd_wg workgroups per member, small
// lattices): the dispatcher deals workgroups round-robin over the eight XCDs (tools/mock/handoff_scope_mock.hip: workgroup b on XCD b % 8,
// 256 of 256), so member  8 * mi + x  takes workgroups  (mi * xcd_wg + j) * 8 + x,  j < xcd_wg -- all on XCD x: its hand-offs stay under one L2
// (0.34 against 0.45 us per hand-off).  Only the latency depends on the placement really being so; the protocol does not.
__device__ __forceinline__ bool persist_wave(const PersistArgs& pa, int& ml, int& w) {
  const int wv = (int)(threadIdx.x >> 6);
  if (pa.xcd_wg > 0) {
    const int xcd = (int)(blockIdx.x & 7), pos = (int)(blockIdx.x >> 3);
    const int mi = pos / pa.xcd_wg, j = pos - mi * pa.xcd_wg;
    ml = __builtin_amdgcn_readfirstlane(mi * 8 + xcd);
    w = __builtin_amdgcn_readfirstlane(j * (kPersistThreads / 64) + wv);
    return ml < pa.nm && w < pa.waves_per_member;
  }
  const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (kPersistThreads / 64)) + wv);
  if (wave >= pa.waves_per_member * pa.nm) return false;
  ml = wave / pa.waves_per_member;
  w = wave - ml * pa.waves_per_member;
  return true;
}

// lane -> (slot, block, node) of wave w of a member (lane_pos of dfx_kernels.h without the workgroup arithmetic)
template <int NPB>
__device__ __forceinline__ LanePos wave_lane_pos(int w, int n_blocks) {
  LanePos p;
  const int gtid = w * 64 + (int)(threadIdx.x & 63);
  if (NPB == 4) { p.slot = gtid; p.b = gtid >> 2; p.k = gtid & 3; p.valid = p.b < n_blocks; return p; }
  const int j = gtid & 15;
  const int tri = min((j * 11) >> 5, 4);
  p.k = j - 3 * tri;
  p.b = (gtid >> 4) * 5 + tri;
  p.valid = p.b < n_blocks;
  p.slot = p.b * 4 + p.k;
  return p;
}

// ---- forward: every stage of every step of one segment in one launch --------------------------------------------------------------
//   records  c.rps > 1: stage i of step n also stores its new record (chunks + velocity) into record i + 1 of step n of the checkpoint
//   state    c.rps == 1 && c.traj: the last stage stores the new step state into the checkpoint of step n + 1
//   stages   c.AD: every stage but the last stores its acceleration
//   none of them (forward only): nothing but the ring moves between stages
// The segment starts from record 0 of its first step (records level) or from stage buffer 0, and -- unless the records level keeps the
// state in the checkpoint -- leaves its last state in stage buffer 0, where k_snapshot and the next segment read it.
#ifndef DFX_PERSIST_OCC
#define DFX_PERSIST_OCC
#endif
template <int MODEL, int CONTACT, int NPB>
__global__ __launch_bounds__(kPersistThreads) DFX_PERSIST_OCC void k_fwd_persist(DevCtx c, PersistCoef pc, PersistArgs pa) {
  int ml, w;
  if (!persist_wave(pa, ml, w)) return;
  const int m = c.m0 + ml;
  const LanePos lp = wave_lane_pos<NPB>(w, c.n_blocks);
  if (!lp.valid) return;
  const int slot = lp.slot, b = lp.b, k = lp.k, kd = k < 3 ? k : 2;
  const u32 nd = (u32)c.n_blocks * 3;
  const int s = c.s;
  const Seg sg = *c.cur;
  const MemberBases B = member_bases(c, m);
  LigRes g;
  load_lig_res<CONTACT>(c, B, slot, g);
  const int info = g.info, pslot = g.pslot;
  // ---- resident per DOF lane
  const int dof = b * 3 + kd;
  const u32 o_dof = (u32)dof * 8, o_rec = ((u32)b * kPos + kd) * 8, o_chunk = ((u32)b * kPos + 2 * (k & 1)) * 8;
  const double damp = c.damping_uniform ? B.cst[6 + kd] : ldg<double>(c.damping + (size_t)((u32)m * nd), o_dof);
  const double invm = ldg<double>(c.inv_m + (size_t)((u32)m * nd), o_dof);
  const int sidx = ldg<int>(c.block_special, (u32)b * 4);
  const bool constrained = k < 3 && sidx >= 0 && ((c.special[sidx >= 0 ? sidx : 0].con_mask >> k) & 1);
  // a driven / loaded DOF's coefficients of the <= 2 time functions, resident (prescribed motion: con_coef, load: load_coef -- a DOF is one or
  // the other); their table values are fetched in FRONT of the poll.  The wave that owns the driven edge would otherwise add two dependent
  // round trips to every stage of its own, and over a segment every wave runs at the pace of the slowest (profiles/r06_persistent_phase_timing.txt)
  double tf_coef[DFX_MAX_FNS];
#pragma unroll
  for (int f = 0; f < DFX_MAX_FNS; ++f)
    tf_coef[f] = (k < 3 && sidx >= 0 && f < c.n_fns) ? (constrained ? c.special[sidx].con_coef[k][f] : c.special[sidx].load_coef[k][f]) : 0.0;
  const bool recs = c.rps > 1;
  const long long n0 = sg.base_step;
  double qn, vn;
  BlockRec<double> o;
  {
    const double* P0 = pos_in(c, m, recs ? -1 : 0, n0);
    const double* V0 = vel_in(c, m, recs ? -1 : 0, n0);
    const double2 a0 = ldg<double2>(P0, (u32)b * kPos * 8), a1 = ldg<double2>(P0, (u32)b * kPos * 8 + 16);
    o.x = a0.x; o.y = a0.y; o.th = a1.x; o.sh = a1.y;
    qn = ldg<double>(P0, o_rec);
    vn = ldg<double>(V0, o_dof);
  }
  double al[kPersistStages];
#pragma unroll
  for (int l = 0; l < kPersistStages; ++l) al[l] = 0.0;
  // ---- the ring: per-lane byte offsets inside a place; a place of all members is ring_stride doubles
  const size_t ring_stride = (size_t)c.batch * c.n_blocks * kPos;
  const u32 r_own = ((u32)m * (u32)c.n_blocks + (u32)b) * (kPos * 8) + 16u * (u32)(k & 1);
  const u32 r_par = ((u32)m * (u32)c.n_blocks + (u32)(pslot >> 2)) * (kPos * 8);
  if (k < 2) ring_store(pa.ring, r_own, k == 0 ? o.x : o.th, k == 0 ? o.y : o.sh);      // ordinal 0
  const int total = pa.n_steps * s;
  int t_ord = 0;
  double v_i = vn;
#ifdef DFX_PERSIST_TIMING
  unsigned acc_t[6] = {0, 0, 0, 0, 0, 0};
  unsigned long long t_prev = tick();
  const unsigned long long t_first = t_prev, w_first = wall_clock64();
#endif
  for (int j = 0; j < pa.n_steps; ++j) {
    const long long n = n0 + j;
    double h = sg.h;
    if (c.t_steps) { const double* ts = steps_of(c, m); h = ts[n + 1] - ts[n]; }
    double* Am = c.AD ? c.AD + (size_t)m * c.ad_stride + (size_t)n * ((u32)(s - 1) * nd) : nullptr;
    // (one copy of the stage body, the stage index a run-time value: unrolled six times the register allocator kept 200 registers live
    // and the launch lost its second and third wave per SIMD)
#pragma unroll 1
    for (int i = 0; i < s; ++i) {
      {
        // ---- what does not need the partner's record, in front of the poll (a wave alone on its SIMD has nothing else to do while it
        // waits): the own half-angle cosine, the Runge-Kutta sums over the EARLIER stages' accelerations, the load of a loaded block
        o.ch = half_cos(o.th, o.sh);
        double sv = 0.0, sq = 0.0, fload = 0.0;
#pragma unroll
        for (int l = 0; l < kPersistStages - 1; ++l) {
          const double a_l = l < i ? al[l] : 0.0;         // (the stage kernels add exact zeros for l >= i as well)
          sv += pc.cv[i][l] * a_l;
          sq += pc.cq[i][l] * a_l;
        }
        // the time functions' table values a driven / loaded DOF needs (prescribed value and rate at the NEXT stage time, or the load at this
        // one): ASKED FOR here, used after the poll -- the two round trips overlap
        double tv0[DFX_MAX_FNS], tv1[DFX_MAX_FNS];
#pragma unroll
        for (int f = 0; f < DFX_MAX_FNS; ++f) { tv0[f] = 0.0; tv1[f] = 0.0; }
        if (k < 3 && sidx >= 0) {
          const double* ft = fn_tab_row(c, m, j, constrained ? i + 1 : i);
          const u32 z = lane_zero();
#pragma unroll
          for (int f = 0; f < DFX_MAX_FNS; ++f)
            if (f < c.n_fns) { tv0[f] = fn_tab_get(ft, f, 0, z); if (constrained) tv1[f] = fn_tab_get(ft, f, 1, z); }
        }
        DFX_TICK(0)
        // ---- the partner's record of this stage
        const double* place = pa.ring + (size_t)(t_ord % kPRing) * ring_stride;
        double pr[4];
        if (!ring_wait(place, r_par, pr, t_ord, pa.give_up, pa.spin_limit)) return;
        DFX_TICK(1)
        if (k < 2 && t_ord + kPAhead <= total) ring_poison(pa.ring + (size_t)((t_ord + kPAhead) % kPRing) * ring_stride, r_own);
        BlockRec<double> p;
        p.x = pr[0]; p.y = pr[1]; p.th = pr[2]; p.sh = pr[3];
        p.ch = half_cos(p.th, p.sh);
        double q_pre = 0.0, v_pre = 0.0;
#pragma unroll
        for (int f = 0; f < DFX_MAX_FNS; ++f) {
          if (constrained) { q_pre += tf_coef[f] * tv0[f]; v_pre += tf_coef[f] * tv1[f]; }
          else fload += tf_coef[f] * tv0[f];
        }
        // ---- ligament + contact of this slot (k_fwd_stage's arithmetic)
        double fx = 0.0, fy = 0.0, fth = 0.0;
        if (info >= 0) {
          BondGrad<double> bg;
          bond_grad<MODEL, double>(o, p, g.rox, g.roy, g.rpx, g.rpy, g.lx, g.ly, g.l0, g.il0, g.ks, g.ksh, g.kr, g.sgn, bg);
          fx = bg.fx; fy = bg.fy; fth = bg.fth;
          if (CONTACT == 1) {
            // the stage kernels fetch a ligament's own void angles only beyond the culling bound and evaluate the member's smallest
            // one otherwise (exact zeros either way); the same selection keeps the two forms equal bit for bit
            const bool far = !(fabs(o.th - p.th) <= g.kap_safe);
            ContactGrad<double> cg;
            contact_grad<double>(g.sgn * (o.th - p.th), far ? g.phi1 : g.phi_min, far ? g.phi2 : g.phi_min, g.am, g.ac, g.kc, cg);
            fth += g.sgn * cg.dkap;
          }
        }
        DFX_TICK(2)
        const double dE = blk_reduce3<NPB>(fx, fy, fth, k);
        // ---- DOF epilogue
        double qnext = 0.0, vnext = 0.0;
        if (k < 3) {
          const double a = constrained ? 0.0 : (fload - dE - damp * v_i) * invm;
          if (Am && i < s - 1) stg<double>(Am + (size_t)i * nd, o_dof, a);
          // (the stage kernels' order: the earlier stages' terms first -- summed above --, the own term last)
          sv += pc.cv[i][i] * a;
          sq += pc.cq[i][i] * a;
#pragma unroll
          for (int l = 0; l < kPersistStages; ++l) al[l] = l == i ? a : al[l];
          qnext = qn + h * (pc.c[i + 1] * vn + h * sq);
          vnext = vn + h * sv;
          if (constrained) { qnext = q_pre; vnext = v_pre; }
        }
        DFX_TICK(3)
        // ---- the next stage record: into the ring for the neighbours, into the checkpoint for the reverse sweep
        const double y1 = blk_bcast<NPB, 1>(qnext, k), th2 = blk_bcast<NPB, 2>(qnext, k), x0 = blk_bcast<NPB, 0>(qnext, k);
        double sn, cs;
        fast_sincos(0.5 * th2, &sn, &cs);
        o.x = x0; o.y = y1; o.th = th2; o.sh = sn;
        ++t_ord;
        DFX_TICK(4)
        const bool last = i == s - 1;
        if (k < 2) ring_store(pa.ring + (size_t)(t_ord % kPRing) * ring_stride, r_own, k == 0 ? x0 : th2, k == 0 ? y1 : sn);
        if (k < 3) {
          const double2 chunk = k == 0 ? make_double2(x0, y1) : make_double2(th2, sn);
          if (recs || (last && c.traj)) {
            double* tr = recs ? traj_rec(c, m, -2 - i, n) : traj_rec(c, m, -1, n + 1);
            if (k < 2) stg<double2>(tr, o_chunk, chunk);
            stg<double>(tr + (size_t)c.n_blocks * kPos, o_dof, vnext);
          }
          if (last && !recs && j == pa.n_steps - 1) {      // the segment's last state, for k_snapshot and the next segment
            if (k < 2) stg<double2>(c.POS + (size_t)((u32)m * (u32)c.nbuf * (u32)c.n_blocks * kPos), o_chunk, chunk);
            stg<double>(c.VEL + (size_t)((u32)m * (u32)c.nbuf * nd), o_dof, vnext);
          }
        }
        v_i = vnext;
        if (last) { qn = qnext; vn = vnext; }
        DFX_TICK(5)
      }
    }
  }
#ifdef DFX_PERSIST_TIMING
  if (pa.dbg && (threadIdx.x & 63) == 0) {
    unsigned* d = pa.dbg + (size_t)(blockIdx.x * 4 + (threadIdx.x >> 6)) * 8;
    for (int q = 0; q < 6; ++q) d[q] = acc_t[q];
    d[6] = (unsigned)(tick() - t_first); d[7] = (unsigned)total;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    for (int q = 0; q < 6; ++q) pa.give_up[1 + q] = (int)acc_t[q];
    pa.give_up[7] = (int)(tick() - t_first);
    pa.give_up[8] = (int)(wall_clock64() - w_first);       // 100 MHz
    pa.give_up[9] = total;
  }
#endif
}


// ---- reverse: every stage of every step of one segment in one launch --------------------------------------------------------------
// The records build of k_adj_stage (records or segments checkpoint, no per-ligament gradients) with everything a DOF or a slot owns
// kept on the compute unit across the segment: lambda and the ligament's parameters in registers, the later stages' Ybar and the
// node-vector / void-angle / inertia / damping accumulators in lane-private LDS (one read-modify-write of the arrays per SEGMENT
// instead of one per stage: 176 of the launch's 611 B per unit).  What
// crosses waves is w = Kbar_v / m of the next stage to run, three doubles per block: the ring record is (w_x, w_y | w_theta, 0).
// The stage records the sweep linearises about come from the checkpoint (plain loads: written by an earlier launch), issued in front
// of the poll so that the two round trips overlap.  The first stage of a launch reads the partner's w where the previous launch
// (k_adj_begin, a stage launch, or this kernel) left it -- DevCtx::W -- and the last one leaves its own there, with lambda in LAM and
// the accumulators in their arrays: segments run by this kernel and by stage launches can alternate.

// (three workgroups per compute unit: 168 registers; the allocator's own choice is 170-172, the limit costs two 8-byte spills of
// epilogue pointers OUTSIDE the stage loop)
//   DENSE: the sweep of an adaptive solve that kept its accepted steps (dfx_dense.h): every member its own number of steps N_m -- a wave
//   runs the steps n <= N_m of the segment, of step N_m only stage 0 (the evaluation at the final state, a step of size zero) -- and the
//   outputs' cotangents enter through the dense output (the same sums as in adj_stage_body<..., DENSE = 1>).  Builds of their own
//   (k_adj_dense_loop, dfx_persist_dense.hip): the fixed-grid kernels carry none of it.
template <int MODEL, int CONTACT, int NPB, int DENSE>
__device__ __forceinline__ void adj_persist_body(const DevCtx& c, const PersistAdjCoef& pc, const PersistArgs& pa, const DenseCtx& dn) {
  int ml, w;
  if (!persist_wave(pa, ml, w)) return;
  const int m = c.m0 + ml;
  const LanePos lp = wave_lane_pos<NPB>(w, c.n_blocks);
  if (!lp.valid) return;
  const int slot = lp.slot, b = lp.b, k = lp.k, kd = k < 3 ? k : 2;
  const u32 nd = (u32)c.n_blocks * 3, nd6 = (u32)c.n_blocks * 6;
  const int s = c.s;
  const Seg sg = *c.cur;
  long long n_m = 0;
  int j_top = pa.n_steps - 1;
  if constexpr (DENSE) {
    n_m = dn.n_acc[m];
    if (n_m - sg.base_step < (long long)j_top) j_top = (int)(n_m - sg.base_step);
    if (j_top < 0) return;                      // this member's sweep starts in an earlier segment
  }
  const MemberBases B = member_bases(c, m);
  LigRes g;
  load_lig_res<CONTACT>(c, B, slot, g);
  const int info = g.info, pslot = g.pslot;
  const int dof = b * 3 + kd;
  const u32 o_dof = (u32)dof * 8, o_b6 = ((u32)b * 6 + 2 * kd) * 8;
  const double damp = c.damping_uniform ? B.cst[6 + kd] : ldg<double>(c.damping + (size_t)((u32)m * nd), o_dof);
  const double invm = ldg<double>(c.inv_m + (size_t)((u32)m * nd), o_dof);
  const int sidx = ldg<int>(c.block_special, (u32)b * 4);
  const bool constrained = k < 3 && sidx >= 0 && ((c.special[sidx >= 0 ? sidx : 0].con_mask >> k) & 1);
  double tf_coef[DFX_MAX_FNS];      // k_fwd_persist: a driven / loaded DOF's coefficients of the time functions, resident
#pragma unroll
  for (int f = 0; f < DFX_MAX_FNS; ++f)
    tf_coef[f] = (k < 3 && sidx >= 0 && f < c.n_fns) ? (constrained ? c.special[sidx].con_coef[k][f] : c.special[sidx].load_coef[k][f]) : 0.0;
  // ---- resident: lambda, accumulators
  double* LAMm = c.LAM + (size_t)((u32)m * nd6);
  const u32 ms = (u32)m * (u32)c.n_slots;
  double* grm = c.g_r + (size_t)(ms * 2);
  double* gpm = c.g_phi + (size_t)ms;
  double* bmm = c.blk_m + (size_t)((u32)m * nd);
  double* bcm = c.blk_c ? c.blk_c + (size_t)((u32)m * nd) : nullptr;
  // The later stages' Ybar (five (q, v) pairs per DOF) and the four accumulators are touched once or twice per stage: they live in LDS,
  // one private place per lane (no lane reads another lane's: no barrier, no fence) -- 30 registers less, which is what lets three
  // workgroups of this kernel share a compute unit (171-198 VGPRs before, two workgroups)
  __shared__ double2 s_yb[kPersistStages - 1][kPersistThreads];
  __shared__ double2 s_racc[kPersistThreads];
  __shared__ double s_acc[3][kPersistThreads];        // void angle, inertia, damping
  __shared__ double2 s_lam[kPersistThreads];
  const int tid = (int)threadIdx.x;
  s_racc[tid] = ldg<double2>(grm, (u32)slot * 16);
  s_lam[tid] = ldg<double2>(LAMm, o_b6);
  s_acc[0][tid] = CONTACT == 1 ? ldg<double>(gpm, (u32)slot * 8) : 0.0;
  s_acc[1][tid] = ldg<double>(bmm, o_dof);
  s_acc[2][tid] = bcm ? ldg<double>(bcm, o_dof) : 0.0;
  bool phi_any = false;
  // ---- the ring
  const size_t ring_stride = (size_t)c.batch * c.n_blocks * kPos;
  const u32 r_own = ((u32)m * (u32)c.n_blocks + (u32)b) * (kPos * 8) + 16u * (u32)(k & 1);
  const u32 r_par = ((u32)m * (u32)c.n_blocks + (u32)(pslot >> 2)) * (kPos * 8);
  const int total = DENSE ? (j_top + 1) * s - ((sg.base_step + j_top == n_m) ? s - 1 : 0) : pa.n_steps * s;
  int t_ord = 0;
#ifdef DFX_PERSIST_TIMING
  unsigned acc_t[6] = {0, 0, 0, 0, 0, 0};
  unsigned long long t_prev = tick();
  const unsigned long long t_first = t_prev;
#endif
  for (int j = j_top; j >= 0; --j) {
    const long long n = sg.base_step + j;
    double h = sg.h, h_before = (sg.j0 + j) == 0 ? sg.h_prev : sg.h;
    if (c.t_steps) { const double* ts = steps_of(c, m); h = ts[n + 1] - ts[n]; h_before = n > 0 ? ts[n] - ts[n - 1] : 0.0; }
#pragma unroll 1
    for (int i = (DENSE && n == n_m) ? 0 : s - 1; i >= 0; --i) {
      const int win = (int)((n * s + i) & 1);
      // ---- the records this stage linearises about (checkpoint), in flight while the partner's w is polled
      const double* POSin = traj_rec(c, m, -1 - i, n);
      const double2 o0 = ldg<double2>(POSin, (u32)b * (kPos * 8)), o1 = ldg<double2>(POSin, (u32)b * (kPos * 8) + 16);
      const double2 q0 = ldg<double2>(POSin, (u32)(pslot >> 2) * (kPos * 8)), q1 = ldg<double2>(POSin, (u32)(pslot >> 2) * (kPos * 8) + 16);
      const double v_i = ldg<double>(POSin + (size_t)c.n_blocks * kPos, o_dof);
      // ---- Kbar sums over the later stages' Ybar, the own w and its broadcasts (k_adj_stage, records build): none of it needs the
      // partner's w or the records still in flight -- in front of the poll
      double sq = 0.0, sv = 0.0, sqc = 0.0, svc = 0.0;
#pragma unroll
      for (int jj = 1; jj < kPersistStages; ++jj) {
        // (DENSE: the evaluation at the final state, stage 0 of the zero-size step N_m, has no later stages -- their places hold nothing yet)
        const bool on = jj > i && jj < s && !(DENSE && n == n_m);
        const double2 y = on ? s_yb[jj - 1][tid] : make_double2(0.0, 0.0);
        const double cf = i > 0 ? pc.col[i][jj] : 1.0;
        sq += cf * y.x;
        sv += cf * y.y;
        sqc += pc.cur[i][jj] * y.x;
        svc += pc.cur[i][jj] * y.y;
      }
      // the load of a loaded DOF at this stage time (tabulated per segment): fetched here, in front of the poll
      double tv0[DFX_MAX_FNS];
#pragma unroll
      for (int f = 0; f < DFX_MAX_FNS; ++f) tv0[f] = 0.0;
      if (k < 3 && sidx >= 0 && !constrained) {
        const double* ft = fn_tab_row(c, m, j, i);
        const u32 z = lane_zero();
#pragma unroll
        for (int f = 0; f < DFX_MAX_FNS; ++f) if (f < c.n_fns) tv0[f] = fn_tab_get(ft, f, 0, z);
      }
      const double col_s = pc.col[i][s], cur_s = pc.cur[i][s], col_i = pc.col[i][i];
      const double2 lam = s_lam[tid];
      const double lq = lam.x, lv = lam.y;
      double w_d = (h * (cur_s * lv + svc)) * invm;
      // dense output: what the outputs inside this step and inside the previous one add to this stage's own Kbar, to the Kbar handed on,
      // and to lambda_n -- scaled by the step sizes, zero on constrained DOFs (adj_stage_body<..., DENSE = 1>, same sums)
      double e_own_q = 0.0, e_nxt_v = 0.0, gs_q = 0.0, gs_v = 0.0;
      if constexpr (DENSE) {
        if (!constrained) {
          const int* op = dn.out_ptr + (size_t)m * dn.stride;
          const int lo = op[n], hi = op[n + 1], plo = n > 0 ? op[n - 1] : lo;
          const double* dwm = dn.dw + (size_t)m * dn.n_out * 8;
          const u32 o_g = ((u32)b * 6 + kd) * 8;
          double e_own_v = 0.0;
          for (int kk = lo; kk < hi; ++kk) {
            const double* Gk = c.G + ((size_t)kk * c.batch + m) * (size_t)nd6;
            const double gq = ldg<double>(Gk, o_g), gv = ldg<double>(Gk, o_g + 24);
            const double* wt = dwm + (size_t)kk * 8;
            e_own_q += wt[i] * gq; e_own_v += wt[i] * gv;
            if (i > 0) e_nxt_v += wt[i - 1] * gv;
            else { gs_q += gq; gs_v += gv; }
          }
          e_own_q *= h; e_own_v *= h; e_nxt_v *= h;
          if (i <= 1) {
            double e6q = 0.0, e6v = 0.0, e5v = 0.0;
            for (int kk = plo; kk < lo; ++kk) {
              const double* Gk = c.G + ((size_t)kk * c.batch + m) * (size_t)nd6;
              const double gq = ldg<double>(Gk, o_g), gv = ldg<double>(Gk, o_g + 24);
              const double* wt = dwm + (size_t)kk * 8;
              e6q += wt[6] * gq; e6v += wt[6] * gv;
              e5v += wt[s - 1] * gv;
            }
            if (i == 0) { e_own_q += h_before * e6q; e_own_v += h_before * e6v; e_nxt_v = h_before * e5v; }
            else e_nxt_v += h_before * e6v;
          }
          if (i == 0 && n == 0)
            for (int kk = 0; kk < lo; ++kk) {
              const double* Gk = c.G + ((size_t)kk * c.batch + m) * (size_t)nd6;
              gs_q += ldg<double>(Gk, o_g); gs_v += ldg<double>(Gk, o_g + 24);
            }
          w_d += e_own_v * invm;
        }
      }
      const double wox = blk_bcast<NPB, 0>(w_d, k), woy = blk_bcast<NPB, 1>(w_d, k), woth = blk_bcast<NPB, 2>(w_d, k);
      double wp[4];
      DFX_TICK(0)
      if (t_ord == 0) {
        const double* Win = c.W + (size_t)(((u32)m * 2 + (u32)win) * nd);
        const u32 pb = (u32)(pslot >> 2) * 24;
        const double2 wxy = ldg<double2>(Win, pb);
        wp[0] = wxy.x; wp[1] = wxy.y; wp[2] = ldg<double>(Win, pb + 16);
      } else if (!ring_wait(pa.ring + (size_t)(t_ord % kPRing) * ring_stride, r_par, wp, t_ord, pa.give_up, pa.spin_limit)) return;
      DFX_TICK(1)
      if (k < 2 && t_ord + kPAhead < total) ring_poison(pa.ring + (size_t)((t_ord + kPAhead) % kPRing) * ring_stride, r_own);
      BlockRec<double> o, p;
      o.x = o0.x; o.y = o0.y; o.th = o1.x; o.sh = o1.y; o.ch = half_cos(o.th, o.sh);
      p.x = q0.x; p.y = q0.y; p.th = q1.x; p.sh = q1.y; p.ch = half_cos(p.th, p.sh);
      // ---- Hessian-vector product + mixed parameter derivatives of this slot
      double hx = 0.0, hy = 0.0, hth = 0.0, ex = 0.0, ey = 0.0, eth = 0.0, d_rx = 0.0, d_ry = 0.0, d_phi = 0.0;
      if (info >= 0) {
        // (the Hessian-vector product written out by hand, as in the records build of the stage kernel: bond_hvp / contact_hvp, dfx_physics.h)
        BondHvp hv;
        bond_hvp<MODEL>(o, p, wox, woy, woth, wp[0], wp[1], wp[2], g.rox, g.roy, g.rpx, g.rpy, g.lx, g.ly, g.l0, g.il0, g.ks, g.ksh, g.kr, g.sgn, hv);
        hx = hv.hx; hy = hv.hy; hth = hv.hth;
        ex = hv.fx; ey = hv.fy; eth = hv.fth;
        d_rx = hv.rx; d_ry = hv.ry;
        if (CONTACT == 1) {
          const bool far = !(fabs(o.th - p.th) <= g.kap_safe);
          double dk, dke, p1e, p2e;
          contact_hvp(g.sgn * (o.th - p.th), g.sgn * (woth - wp[2]), far ? g.phi1 : g.phi_min, far ? g.phi2 : g.phi_min, g.am, g.ac, g.kc, dk, dke, p1e, p2e);
          hth += g.sgn * dke;
          eth += g.sgn * dk;
          d_phi = (info & 1) ? p2e : p1e;
        }
        const double2 r_old = s_racc[tid];
        s_racc[tid] = make_double2(r_old.x - d_rx, r_old.y - d_ry);
      }
      if (CONTACT == 1 && d_phi != 0.0) { s_acc[0][tid] -= d_phi; phi_any = true; }
      DFX_TICK(2)
      const double hw = blk_reduce3<NPB>(hx, hy, hth, k);
      const double dE = blk_reduce3<NPB>(ex, ey, eth, k);
      // ---- DOF epilogue
      double w_next = 0.0;
      if (k < 3) {
        double fload = 0.0;
#pragma unroll
        for (int f = 0; f < DFX_MAX_FNS; ++f) if (!constrained) fload += tf_coef[f] * tv0[f];
        if (c.fn_g && sidx >= 0) {        // gradients w.r.t. the time functions' parameters (asked for explicitly: not the design loop's path)
          const double* ft = fn_tab_row(c, m, j, i);
#pragma unroll
          for (int f = 0; f < DFX_MAX_FNS; ++f) {
            const double coef = f < c.n_fns ? (constrained ? -hw * tf_coef[f] : w_d * tf_coef[f]) : 0.0;
            if (coef != 0.0) {
              const u32 z = lane_zero();
              double* q = c.fn_g + (((size_t)m * c.n_special + sidx) * DFX_MAX_FNS + f) * DFX_FN_PARAMS;
              for (int kk = 0; kk < DFX_FN_PARAMS; ++kk) acc_add(q + kk, coef * fn_tab_get(ft, f, 2 + kk, z));
            }
          }
        }
        const double a_i = constrained ? 0.0 : (fload - dE - damp * v_i) * invm;
        double kq_in = h * (cur_s * lq + sqc);
        if constexpr (DENSE) kq_in += e_own_q;
        double ybq = 0.0, ybv = 0.0;
        if (!constrained) {
          ybq = -hw;
          ybv = kq_in - damp * w_d;
          s_acc[1][tid] -= w_d * a_i;
          s_acc[2][tid] -= w_d * v_i;
        }
        if (i > 0) s_yb[i - 1][tid] = make_double2(ybq, ybv);
        double kv;
        if (i > 0) {
          kv = h * (col_s * lv + col_i * ybv + sv);
        } else {
          double nlq = lq + (ybq + sq), nlv = lv + (ybv + sv);
          if constexpr (DENSE) { nlq += gs_q; nlv += gs_v; }
          const bool first = !DENSE && (sg.j0 + j) == 0;
          if (first && c.G && !constrained) {
            const double* G = c.G + ((size_t)sg.interval * c.batch + m) * (size_t)nd6;
            nlq += G[b * 6 + k]; nlv += G[b * 6 + 3 + k];
          }
          if (constrained) { nlq = 0.0; nlv = 0.0; }
          s_lam[tid] = make_double2(nlq, nlv);
          kv = h_before * col_s * nlv;
        }
        if constexpr (DENSE) kv += e_nxt_v;
        w_next = constrained ? 0.0 : kv * invm;
      }
      DFX_TICK(3)
      // ---- w of the next stage to run: into the ring, or -- last stage of the launch -- where the next launch reads it
      ++t_ord;
      if (t_ord < total) {
        const double w1 = blk_bcast<NPB, 1>(w_next, k), w2 = blk_bcast<NPB, 2>(w_next, k);
        if (k < 2) ring_store(pa.ring + (size_t)(t_ord % kPRing) * ring_stride, r_own, k == 0 ? w_next : w2, k == 0 ? w1 : 0.0);
      } else if (k < 3) {
        stg<double>(c.W + (size_t)(((u32)m * 2 + (u32)(win ^ 1)) * nd), o_dof, w_next);
      }
    }
  }
#ifdef DFX_PERSIST_TIMING
  if (pa.dbg && (threadIdx.x & 63) == 0) {
    unsigned* d = pa.dbg + (size_t)(blockIdx.x * 4 + (threadIdx.x >> 6)) * 8;
    for (int q = 0; q < 6; ++q) d[q] = acc_t[q];
    d[6] = (unsigned)(tick() - t_first); d[7] = (unsigned)total;
  }
#endif
  // ---- what the segment leaves behind
  if (info >= 0) stg<double2>(grm, (u32)slot * 16, s_racc[tid]);
  if (CONTACT == 1 && phi_any) { stg<double>(gpm, (u32)slot * 8, s_acc[0][tid]); c.touch[0] = 1; }
  if (k < 3) {
    stg<double2>(LAMm, o_b6, s_lam[tid]);
    if (!constrained) { stg<double>(bmm, o_dof, s_acc[1][tid]); if (bcm) stg<double>(bcm, o_dof, s_acc[2][tid]); }
  }
}

template <int MODEL, int CONTACT, int NPB>
__global__ __launch_bounds__(kPersistThreads) __attribute__((amdgpu_waves_per_eu(3))) void k_adj_persist(DevCtx c, PersistAdjCoef pc, PersistArgs pa) {
  adj_persist_body<MODEL, CONTACT, NPB, 0>(c, pc, pa, DenseCtx{});
}

}  // namespace
