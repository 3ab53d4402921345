// dfx_engine.hip -- libdfx: the MI355X (gfx950) engine behind include/dfx.h.
//
// Execution model
//   * one lane per (block, node slot): 4 lanes = one rigid unit, 16 units per 64-wide wavefront,
//     256-thread workgroups, grid = (ceil(4*n_blocks/256), batch members).  Every ligament is evaluated
//     by both of its end lanes ("gather form"); the 4 slot contributions of a unit are summed with
//     quad shuffles; lanes 0..2 of the quad then own DOF x, y, theta for the integrator epilogue.
//     No atomics anywhere: results are bit-reproducible.
//   * one kernel launch per Runge-Kutta stage (the neighbour exchange of an explicit stage is a grid-wide
//     dependency; a kernel boundary is the cheapest grid barrier on this chip).
//   * a launch never chases an index: the epilogue of stage i assembles the stage record of stage i+1
//     (64 B per unit: x y th cos(th/2) sin(th/2) vx vy vth) AND pushes the 5 numbers a neighbour needs
//     into that neighbour's "mailbox" slot, so every load address of the next launch is known at wave
//     start (one latency level instead of index -> gather).  All parameter tables are 16-byte rows
//     indexed by the lane's own slot (coalesced dwordx4).
//   * the time loop is replayed from hipGraphs (one graph = one segment of <= kMaxGraphSteps steps);
//     what changes between replays (time, step size, checkpoint slot) is one 64-byte record in device
//     memory, refreshed by a 1-thread tick kernel at the head of each graph.
//   * the reverse sweep re-reads the checkpointed trajectory (one 64-B record per unit per step, kept in
//     HBM), recomputes the stage records of a step and runs one Dual-number kernel per stage.
//
// The per-ligament physics (dfx_physics.h) is shared with the CPU port of the oracle.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "dfx_stage.h"

using namespace dfx;

// occupancy hints for the two stage kernels (waves per SIMD the register allocator must leave room for).
// Forward: 5 waves (96 VGPRs + 100 B/lane of scratch) measured +7 % forward-only at 16 members per GPU over the
// allocator's own choice (120 VGPRs, 4 waves), neutral at 1..8 members; 6 waves lose it again to spills.  Reverse: any
// forced occupancy spills heavily (-25 %), left to the allocator (150 VGPRs, 3 waves).
#ifndef DFX_FWD_OCC
#define DFX_FWD_OCC __attribute__((amdgpu_waves_per_eu(5)))
#endif
#ifndef DFX_ADJ_OCC
#define DFX_ADJ_OCC
#endif

namespace {

// Index arithmetic inside the stage kernels is 32-bit (one s_mul / v_mad instead of a 64-bit multiply chain per array);
// dfx_create refuses ensembles whose largest per-handle array would not fit (check_index_range).  The trajectory
// checkpoint and the cotangent table keep 64-bit offsets.
typedef unsigned u32;

constexpr int kThreads = 256;
constexpr int kAccCap = 1 << 20;   // accepted step times recorded per member (adaptive)
constexpr int kMaxGraphSteps = 256;
constexpr int kPos = 6;   // doubles per unit position record: x y th cos(th/2) sin(th/2) pad   (three 16-byte chunks)
constexpr int kStep = 9;  // doubles per unit in a trajectory checkpoint: position record + velocity (3)

struct Seg {            // one graph replay worth of steps
  double t_interval;    // timepoints[k]
  double h;             // step size of the interval
  double h_prev;        // step size of the previous interval (reverse sweep, first step of an interval)
  long long base_step;  // global index of the first step of the segment
  int j0;               // index of that step inside its interval
  int interval;         // k
  int n_steps;
  int pad;
};

// adaptive mode: every member carries its own clock and step-size controller state
struct Clock {
  double t, h;            // start time and size of the step being attempted
  double t_last, h_acc;   // start and size of the last accepted step (dense output)
  long long attempts, accepted;
  int state;              // 0 running, 1 finished, 2 non-finite error estimate, 3 step size underflow
  int fin_next;           // the last output was produced in this round: finished from the next round on
  int accept;             // decision of the last controller run
  int out_lo, out_hi;     // outputs [out_lo, out_hi) lie inside the step just accepted
  int out_idx;            // next output to produce
};

// next-stage coefficients of one launch, passed by value (lands in SGPRs)
struct StageCoef {
  double cv[kMaxStages];  // a[r][l]   : V_{r} = v_n + h sum_l cv[l] A_l
  double cq[kMaxStages];  // (a*a)[r][l]: Q_{r} = q_n + h c_r v_n + h^2 sum_l cq[l] A_l
  double c_i, c_next;     // stage times
};

// reverse-stage coefficients: kbar_{i-1} = h (b_{i-1} lambda + sum_{j>=i} a[j][i-1] Ybar_j)
struct AdjCoef {
  double col[kMaxStages + 1];  // col[j] = a[j][i-1] for j in i..s-1, col[s] = b_{i-1};  at i == 0: col[s] = b_{s-1}
  double c_i;
};

struct DevCtx {
  int n_blocks, n_slots, n_fns, batch, s, n_special, k_uniform, n_timepoints;
  int m0, nbuf;           // first member of the group this launch integrates (one stream per group); stage buffers per member
  int pred[4];            // guessed partner slot = own slot + pred[node slot]
  int ablate, n_wg;       // DFX_ABLATE: profiling experiments only (results are wrong when non-zero); workgroups per member
  long long traj_stride;  // elements between members in traj
  const int32_t* slot_info;
  const int32_t* block_special;
  const dfx_special* special;
  // per-member parameter images, rows indexed by the lane's own slot / DOF (coalesced)
  const double* p_r;      // n_slots*2   own centroid->node vector
  const double* p_l;      // n_slots*2   reference vector of the slot's ligament
  const double* p_k;      // n_slots*4   stiffnesses (only read when they differ between ligaments)
  const double* p_phi;    // n_slots     undeformed void angle: phi1 on end-0 slots, phi2 on end-1 slots (the other one is gathered from the partner slot)
  const uint8_t* p_lidx;  // n_slots     index of the slot's reference vector in l_dict (when l_dict_on)
  const double* l_dict;   // 256*4   lx ly |l0| 1/|l0|
  int l_dict_on, damping_uniform;
  const double* cst;      // 16           min_angle cutoff_angle k_contact | uniform k_stretch k_shear k_rot
  const double* inv_m;    // n_blocks*3
  const double* damping;  // n_blocks*3
  const TimeFn* fns;
  const Seg* cur;         // the segment being replayed
  Clock* clock;           // per-member clocks (adaptive mode) or null
  double* err_partial;    // batch * n_wg*4 per-wave partial sums of the squared error ratio
  int* step_counts;       // batch * (n_timepoints-1) accepted steps per output interval (adaptive)
  double* acc_times;      // batch * acc_cap end times of the accepted steps (adaptive)
  int acc_cap;
  const double* t_steps;  // n_total+1 step boundaries of a caller-chosen grid, or null: equal steps (Seg.h)
  double* AD;             // stage checkpoint: batch * (N * s * n_dof) stage accelerations of EVERY step, or null
  long long ad_stride;    // elements between members in AD
  const double* ts_dev;   // output times (adaptive mode)
  double* fields_dev;     // batch * T * n_blocks*6 (adaptive mode writes its dense output here)
  double rtol, atol;
  // state: (s+1) stage buffers per member; buffer 0 = current step state
  double* traj;           // batch * traj_stride   checkpoints: per step [POS n_blocks*6 | VEL n_blocks*3]
  double* POS;            // batch * (s+1) * n_blocks*kPos
  double* VEL;            // batch * (s+1) * n_blocks*3
  double* A;              // batch * s * n_blocks*3
  // reverse
  double* YB;             // batch * s * n_blocks*6
  double* LAM;            // batch * n_blocks*6
  double* W;              // batch * 2 * n_blocks*3
  double* KQ;             // batch * 2 * n_blocks*3
  const double* G;        // T * batch * n_blocks*6 (time-major)
  double* g_r;            // batch * n_slots*2     d/d(own node vector)
  double* g_phi;          // batch * n_slots       d/d(void angle): phi1 on the end-0 slot of a ligament, phi2 on its end-1 slot
  double* g_b;            // batch * n_slots*8     d/d(l0(2), k(3), contact(3)) (end-1 slots) or null
  double* blk_m;          // batch * n_blocks*3    d/d(inertia)
  double* blk_c;          // batch * n_blocks*3    d/d(damping) or null
  double* fn_g;           // batch * n_special*MAX_FNS*FN_PARAMS or null
};

// ---- quad (4-lane) data movement on DPP: no LDS traffic, no bank conflicts --------------------
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double quad_sum(double v) {
  v += dpp_mov<0xB1>(v);  // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);  // quad_perm [2,3,0,1]
  return v;
}
// accumulate into an array that several lanes of one launch may hit (time-function parameter gradients of a block whose
// DOFs share a function): hardware fp64 atomic add performed in L2, result not returned
__device__ __forceinline__ void acc_add(double* p, double v) { (void)unsafeAtomicAdd(p, v); }

template <int J>
__device__ __forceinline__ double quad_bcast(double v) { return dpp_mov<J | (J << 2) | (J << 4) | (J << 6)>(v); }

// XCD-aware workgroup order: hardware deals workgroups round-robin over the 8 XCDs (id % 8 share an L2);
// give every XCD one contiguous band of the lattice so neighbour gathers mostly hit that XCD's own L2.
__device__ __forceinline__ int logical_wg(int bid, int n_wg) {
  const int x = bid & 7, q = bid >> 3, per = n_wg >> 3, rem = n_wg & 7;
  return x * per + (x < rem ? x : rem) + q;
}

__device__ __forceinline__ const double* pos_in(const DevCtx& c, int m, int buf, long long n) {
  if (buf >= 0) return c.POS + ((size_t)m * c.nbuf + buf) * (u32)c.n_blocks * kPos;
  return c.traj + (size_t)m * c.traj_stride + (size_t)n * c.n_blocks * kStep;
}
__device__ __forceinline__ const double* vel_in(const DevCtx& c, int m, int buf, long long n) {
  if (buf >= 0) return c.VEL + ((size_t)m * c.nbuf + buf) * (u32)c.n_blocks * 3;
  return c.traj + (size_t)m * c.traj_stride + (size_t)n * c.n_blocks * kStep + (size_t)c.n_blocks * kPos;
}

__global__ void k_tick(const Seg* segs, int* seg_idx, int delta, Seg* cur) {
  int i = *seg_idx + delta;
  *seg_idx = i;
  *cur = segs[i];
}

struct TimeVals { double g, gt; };

// value of the prescribed displacement of DOF d of special block sp at time t (and its rate)
__device__ __forceinline__ TimeVals constrained_value(const DevCtx& c, int m, const dfx_special& sp, int d, double t) {
  TimeVals r{0.0, 0.0};
  double gp[kMaxFnParams];
  for (int f = 0; f < c.n_fns; ++f)
    if (sp.con_coef[d][f] != 0.0) {
      double g, gt;
      eval_time_fn(c.fns[(size_t)m * DFX_MAX_FNS + f], t, g, gt, gp);
      r.g += sp.con_coef[d][f] * g;
      r.gt += sp.con_coef[d][f] * gt;
    }
  return r;
}

// records of a full (2, n_blocks, 3) state at time t0 -> stage buffer `buf`  (constrained DOFs follow c(t0), c'(t0))
__global__ __launch_bounds__(kThreads) void k_init(DevCtx c, const double* state0, double t0, int buf) {
  const int m = blockIdx.y + c.m0;
  const int tid = blockIdx.x * kThreads + threadIdx.x;
  if (tid >= c.n_slots) return;
  const int b = tid >> 2, d = tid & 3;
  if (d == 3) return;
  const size_t nd = (size_t)c.n_blocks * 3;
  double q = state0[(size_t)m * 2 * nd + b * 3 + d], v = state0[(size_t)m * 2 * nd + nd + b * 3 + d];
  const int sidx = c.block_special[b];
  if (sidx >= 0 && ((c.special[sidx].con_mask >> d) & 1)) {
    TimeVals tv = constrained_value(c, m, c.special[sidx], d, t0);
    q = tv.g; v = tv.gt;
  }
  double* pr = c.POS + ((size_t)m * c.nbuf + buf) * c.n_blocks * kPos + (size_t)b * kPos;
  pr[d] = q;
  c.VEL[((size_t)m * c.nbuf + buf) * nd + b * 3 + d] = v;
  if (d == 2) {
    double sn, cs;
    fast_sincos(0.5 * q, &sn, &cs);
    pr[3] = cs; pr[4] = sn; pr[5] = 0.0;
  }
}

// fields[m, k] <- (disp, vel) of stage buffer 0
__global__ __launch_bounds__(kThreads) void k_snapshot(DevCtx c, double* fields, int k, int* bad) {
  const int m = blockIdx.y + c.m0;
  const int tid = blockIdx.x * kThreads + threadIdx.x;
  if (tid >= c.n_blocks * 3) return;
  const int b = tid / 3, d = tid % 3;
  double* f = fields + ((size_t)m * c.n_timepoints + k) * c.n_blocks * 6;
  const double q = c.POS[(size_t)m * c.nbuf * c.n_blocks * kPos + (size_t)b * kPos + d];
  const double v = c.VEL[(size_t)m * c.nbuf * c.n_blocks * 3 + tid];
  f[tid] = q;
  f[(size_t)c.n_blocks * 3 + tid] = v;
  if (!isfinite(q) || !isfinite(v)) *bad = k + 1;   // any writer wins: only "some output row is not finite" matters
}

// copy stage buffer 0 into checkpoint slot n (only for the initial state)
__global__ __launch_bounds__(kThreads) void k_checkpoint0(DevCtx c) {
  const int m = blockIdx.y + c.m0;
  const int tid = blockIdx.x * kThreads + threadIdx.x;
  if (tid >= c.n_blocks * kStep) return;
  double* t = c.traj + (size_t)m * c.traj_stride;
  if (tid < c.n_blocks * kPos) t[tid] = c.POS[(size_t)m * c.nbuf * c.n_blocks * kPos + tid];
  else t[tid] = c.VEL[(size_t)m * c.nbuf * c.n_blocks * 3 + (tid - c.n_blocks * kPos)];
}

// Addressing: every array access in the stage kernels is  uniform base (SGPR pair: kernel argument + member / buffer
// offsets, scalar arithmetic)  +  32-bit per-lane byte offset (one VGPR, shared by all arrays that are indexed the same
// way).  Written this way the compiler emits the `global_load v, v_off, s[base:base+1]` form: no 64-bit vector address
// arithmetic and no VGPR pair per array.
template <class T>
__device__ __forceinline__ T ldg(const void* base, u32 byte_off) {
  return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off);
}
template <class T>
__device__ __forceinline__ void stg(void* base, u32 byte_off, T v) {
  *reinterpret_cast<T*>(reinterpret_cast<char*>(base) + byte_off) = v;
}

// uniform bases of member m's parameter arrays
struct MemberBases {
  const double *p_r, *p_phi, *p_l, *p_k, *cst, *l_dict;
  const uint8_t* p_lidx;
};
__device__ __forceinline__ MemberBases member_bases(const DevCtx& c, int m) {
  MemberBases B;
  const size_t ps = (size_t)m * (u32)c.n_slots;
  B.p_r = c.p_r + ps * 2; B.p_phi = c.p_phi + ps; B.p_l = c.p_l + ps * 2; B.p_k = c.p_k + ps * 4;
  B.cst = c.cst + (size_t)m * 16; B.l_dict = c.l_dict + (size_t)m * 1024; B.p_lidx = c.p_lidx + ps;
  return B;
}

struct LaneIn {
  BlockRec<double> o, p;
  double rox, roy, rpx, rpy, lx, ly, l0, il0, ks, ksh, kr, phi1, phi2, am, ac, kc, sgn;
  int info, pslot, guess;
};

struct Partner {
  double2 b0, b1, rp;
  double b2, phi;
};

template <int CONTACT>
__device__ __forceinline__ void load_partner(const MemberBases& B, int pslot, const double* POSin, Partner& P) {
  const u32 rec = (u32)(pslot >> 2) * (kPos * 8);
  P.b0 = ldg<double2>(POSin, rec);
  P.b1 = ldg<double2>(POSin, rec + 16);
  P.b2 = ldg<double>(POSin, rec + 32);
  P.rp = ldg<double2>(B.p_r, (u32)pslot * 16);
  P.phi = CONTACT ? ldg<double>(B.p_phi, (u32)pslot * 8) : 0.0;
}

// Everything a lane needs for its ligament, in two phases so that every load that does not depend on another
// load is in flight before the first wait:
//   issue_lane   own data (one coalesced 16-byte chunk per lane, spread over the quad by DPP later), slot_info, and
//                the partner's data from a GUESSED slot (own slot + the lattice's usual offset for this node slot);
//                the kernels then issue their own epilogue operands;
//   resolve_lane what depends on loaded values: the dictionary entry of the reference vector, and a second gather
//                only for lanes whose real partner is not the guessed one (irregular connectivity).
// Partner data comes from the same arrays the owners read (lines served by the XCD's L2).
struct LaneRaw {
  Partner P;
  double2 pc, ro, lv;
  double ks, ksh, kr, phi;
  int info, guess, lidx;
};

template <int CONTACT>
__device__ __forceinline__ void issue_lane(const DevCtx& c, const MemberBases& B, int slot, const double* POSin, LaneRaw& R) {
  const int b = slot >> 2, k = slot & 3;
  R.info = ldg<int>(c.slot_info, (u32)slot * 4);
  R.pc = k < 3 ? ldg<double2>(POSin, ((u32)b * kPos + 2 * k) * 8) : make_double2(0.0, 0.0);
  R.ro = ldg<double2>(B.p_r, (u32)slot * 16);
  // branch-free (a branch here would end the batch of loads): the unused one of the two reads one shared valid address
  R.lidx = (int)ldg<uint8_t>(c.l_dict_on ? (const void*)B.p_lidx : (const void*)B.cst, c.l_dict_on ? (u32)slot : 0u);
  R.lv = ldg<double2>(c.l_dict_on ? B.cst : B.p_l, c.l_dict_on ? 0u : (u32)slot * 16);
  R.ks = R.ksh = R.kr = 0.0;
  if (!c.k_uniform) { R.ks = ldg<double>(B.p_k, (u32)slot * 32); R.ksh = ldg<double>(B.p_k, (u32)slot * 32 + 8); R.kr = ldg<double>(B.p_k, (u32)slot * 32 + 16); }
  R.phi = CONTACT ? ldg<double>(B.p_phi, (u32)slot * 8) : 0.0;
  const int delta = k == 0 ? c.pred[0] : (k == 1 ? c.pred[1] : (k == 2 ? c.pred[2] : c.pred[3]));   // selects: a dynamic index would be a memory load
  R.guess = min(max(slot + delta, 0), c.n_slots - 1);
  load_partner<CONTACT>(B, R.guess, POSin, R.P);
}

template <int CONTACT>
__device__ __forceinline__ void resolve_lane(const DevCtx& c, const MemberBases& B, const double* POSin, LaneRaw& R, LaneIn& L) {
  const int info = R.info;
  L.info = info;
  double2 lv = R.lv, ln = make_double2(0.0, 0.0);
  if (c.l_dict_on) {
    lv = ldg<double2>(B.l_dict, (u32)R.lidx * 32);
    ln = ldg<double2>(B.l_dict, (u32)R.lidx * 32 + 16);
  }
  const int pslot = info < 0 ? R.guess : (info >> 1);
  L.pslot = pslot; L.guess = R.guess;
  if (pslot != R.guess) load_partner<CONTACT>(B, pslot, POSin, R.P);
  const double* cst = B.cst;
  if (c.k_uniform) { L.ks = cst[3]; L.ksh = cst[4]; L.kr = cst[5]; }
  else { L.ks = R.ks; L.ksh = R.ksh; L.kr = R.kr; }
  if (CONTACT) {
    L.phi1 = (info & 1) ? R.P.phi : R.phi;
    L.phi2 = (info & 1) ? R.phi : R.P.phi;
    L.am = cst[0]; L.ac = cst[1]; L.kc = cst[2];
  }
  L.o.x = quad_bcast<0>(R.pc.x); L.o.y = quad_bcast<0>(R.pc.y);
  L.o.th = quad_bcast<1>(R.pc.x); L.o.ch = quad_bcast<1>(R.pc.y);
  L.o.sh = quad_bcast<2>(R.pc.x);
  L.p.x = R.P.b0.x; L.p.y = R.P.b0.y; L.p.th = R.P.b1.x; L.p.ch = R.P.b1.y; L.p.sh = R.P.b2;
  L.rox = R.ro.x; L.roy = R.ro.y; L.rpx = R.P.rp.x; L.rpy = R.P.rp.y;
  L.lx = lv.x; L.ly = lv.y;
  if (c.l_dict_on) { L.l0 = ln.x; L.il0 = ln.y; }
  else {
    L.l0 = info < 0 ? 1.0 : sqrt(lv.x * lv.x + lv.y * lv.y);
    L.il0 = 1.0 / L.l0;
  }
  L.sgn = (info & 1) ? 1.0 : -1.0;
}

template <int CONTACT>
__device__ __forceinline__ void load_lane(const DevCtx& c, int m, int slot, const double* POSin, LaneIn& L) {
  const MemberBases B = member_bases(c, m);
  LaneRaw R;
  issue_lane<CONTACT>(c, B, slot, POSin, R);
  resolve_lane<CONTACT>(c, B, POSin, R, L);
}

// ---- forward stage ---------------------------------------------------------------------------
//   in_buf  : stage buffer holding this stage's records, or -1: the checkpoint of step n (reverse recompute, i == 0)
//   out_buf : buffer for the next stage's records (-1: none)
//   y_buf   : 0: step base state (q_n, v_n) in buffer 0;  -1: in the checkpoint of step n
//   write_traj: also store the new record into the checkpoint of step n+1 (last stage, keep_trajectory)
template <int MODEL, int CONTACT>
__global__ __launch_bounds__(kThreads) DFX_FWD_OCC void k_fwd_stage(DevCtx c, StageCoef sc, int i, int j, int in_buf, int out_buf,
                                                        int y_buf, int mode) {
  const int m = blockIdx.y + c.m0;
  const int lwg = logical_wg(blockIdx.x, c.n_wg);
  int slot = lwg * kThreads + threadIdx.x;
  const int write_traj = mode & 1, err_mode = mode & 2;
  const bool valid = slot < c.n_slots;
  if (!valid) {
    if (!err_mode) return;
    slot = c.n_slots - 4 + (threadIdx.x & 3);   // keep the wave whole for the reduction: redo the last unit, contribute 0
  }
  if (c.ablate & 4) return;
  const int b = slot >> 2, k = slot & 3, kd = k < 3 ? k : 2;
  Seg sg = *c.cur;
  if (c.clock) {          // adaptive: this member's own time and step
    const Clock ck = c.clock[m];
    if (ck.state | ck.fin_next) return;
    sg.t_interval = ck.t; sg.h = ck.h; sg.j0 = 0; sg.base_step = 0; j = 0;
  }
  const long long n = sg.base_step + j;
  const u32 nd = (u32)c.n_blocks * 3;
  // ---- load phase
  const double* POSin = pos_in(c, m, in_buf, n);
  const MemberBases B = member_bases(c, m);
  LaneRaw R;
  issue_lane<CONTACT>(c, B, slot, POSin, R);
  const int dof = b * 3 + kd;
  const u32 o_dof = (u32)dof * 8, o_rec = ((u32)b * kPos + kd) * 8;      // per-lane byte offsets shared by all per-DOF arrays
  const double qn = ldg<double>(pos_in(c, m, y_buf, n), o_rec);
  const double vn = ldg<double>(vel_in(c, m, y_buf, n), o_dof);
  const double v_i = ldg<double>(vel_in(c, m, in_buf, n), o_dof);
  // stage accelerations: the per-member scratch set, or (stage checkpoint) this step's own slot, kept for the reverse sweep
  // (the last stage's acceleration is not kept: no stage record depends on it)
  const bool keep_stages = c.AD && !c.clock;
  double* Am = keep_stages ? c.AD + (size_t)m * c.ad_stride + (size_t)n * ((u32)(c.s - 1) * nd) : c.A + (size_t)m * (u32)(c.s + 1) * nd;
  const double damp = c.damping_uniform ? B.cst[6 + kd] : ldg<double>(c.damping + (size_t)m * nd, o_dof);
  const double invm = ldg<double>(c.inv_m + (size_t)m * nd, o_dof);
  const int sidx = ldg<int>(c.block_special, (u32)b * 4);
  // earlier stage accelerations: all loads issued together (a rolled loop waits for each one in turn)
  double al[kMaxStages - 1];
#pragma unroll
  for (int l = 0; l < kMaxStages - 1; ++l) al[l] = l < i ? ldg<double>(Am + (size_t)l * nd, o_dof) : 0.0;
  LaneIn L;
  resolve_lane<CONTACT>(c, B, POSin, R, L);
  double sv = 0.0, sq = 0.0;
#pragma unroll
  for (int l = 0; l < kMaxStages - 1; ++l) {
    sv += sc.cv[l] * al[l];
    sq += sc.cq[l] * al[l];
  }
  // ---- ligament + contact of this slot
  double fx = 0.0, fy = 0.0, fth = 0.0;
  if (c.ablate & 1) {
    fx = L.o.x + L.p.x + L.rox + L.rpx + L.lx + L.l0; fy = L.o.y + L.p.y + L.roy + L.rpy + L.ly + L.il0;
    fth = L.o.th + L.p.th + L.o.ch + L.p.ch + L.o.sh + L.p.sh + (CONTACT ? L.phi1 + L.phi2 : 0.0);
  } else if (L.info >= 0) {
    BondGrad<double> g;
    bond_grad<MODEL, double>(L.o, L.p, L.rox, L.roy, L.rpx, L.rpy, L.lx, L.ly, L.l0, L.il0, L.ks, L.ksh, L.kr, L.sgn, g);
    fx = g.fx; fy = g.fy; fth = g.fth;
    if (CONTACT) {
      ContactGrad<double> cg;
      contact_grad<double>(L.sgn * (L.o.th - L.p.th), L.phi1, L.phi2, L.am, L.ac, L.kc, cg);
      fth += L.sgn * cg.dkap;
    }
  }
  fx = quad_sum(fx);
  fy = quad_sum(fy);
  fth = quad_sum(fth);
  // ---- DOF epilogue on lanes 0..2
  double h = sg.h, t = sg.t_interval + (sg.j0 + j) * sg.h;
  if (c.t_steps && !c.clock) { t = c.t_steps[n]; h = c.t_steps[n + 1] - t; }
  double qnext = 0.0, vnext = 0.0;
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
            eval_time_fn(c.fns[(u32)m * DFX_MAX_FNS + f], t + sc.c_i * h, g, gt, gp);
            fload += sp.load_coef[k][f] * g;
          }
      }
    }
    const double a = constrained ? 0.0 : (fload - dE - damp * v_i) * invm;
    if (!(keep_stages && i == c.s - 1)) stg<double>(Am + (size_t)i * nd, o_dof, a);
    sv += sc.cv[i] * a;
    sq += sc.cq[i] * a;
    qnext = qn + h * (sc.c_next * vn + h * sq);
    vnext = vn + h * sv;
    if (err_mode) {
      // i == 6 evaluated at the candidate y1: with (cv, cq) = (e, ee) the sums are the embedded error estimate
      double r2 = 0.0;
      if (!constrained) {
        const double q1 = ldg<double>(POSin, o_rec);
        const double eq = h * h * sq, ev = h * sv;
        const double tq = c.atol + c.rtol * fmax(fabs(qn), fabs(q1)), tv = c.atol + c.rtol * fmax(fabs(vn), fabs(v_i));
        r2 = (eq / tq) * (eq / tq) + (ev / tv) * (ev / tv);
      }
      qnext = r2;
    }
    if (constrained && out_buf >= 0) {
      TimeVals tv = constrained_value(c, m, c.special[sidx], k, t + sc.c_next * h);
      qnext = tv.g; vnext = tv.gt;
    }
  }
  if (err_mode) {
    // per-wave sum of the squared error ratios (fixed order -> deterministic); lane 0 of each wave stores it
    double r2 = (k < 3 && valid) ? qnext : 0.0;
    for (int off = 32; off > 0; off >>= 1) r2 += __shfl_down(r2, off, 64);
    if ((threadIdx.x & 63) == 0) c.err_partial[((u32)m * c.n_wg + lwg) * 4 + (threadIdx.x >> 6)] = r2;
    return;
  }
  if (out_buf < 0) return;
  // ---- publish the next stage record: lanes 0..2 each store one aligned 16-byte chunk (x,y) (th,ch) (sh,0)
  const double y1 = quad_bcast<1>(qnext), th2 = quad_bcast<2>(qnext);
  double sn, cs;
  fast_sincos(0.5 * th2, &sn, &cs);
  const double2 chunk = k == 0 ? make_double2(qnext, y1) : (k == 1 ? make_double2(th2, cs) : make_double2(sn, 0.0));
  if (k < 3 && !(c.ablate & 2)) {
    const u32 o_chunk = ((u32)b * kPos + 2 * k) * 8;
    stg<double2>(c.POS + ((size_t)m * c.nbuf + out_buf) * (u32)c.n_blocks * kPos, o_chunk, chunk);
    stg<double>(c.VEL + ((size_t)m * c.nbuf + out_buf) * nd, o_dof, vnext);
    if (write_traj) {
      double* tr = c.traj + (size_t)m * c.traj_stride + (size_t)(n + 1) * c.n_blocks * kStep;
      stg<double2>(tr, o_chunk, chunk);
      stg<double>(tr + (size_t)c.n_blocks * kPos, o_dof, vnext);
    }
  }
}



// ---- adaptive step control (jax.experimental.ode semantics) --------------------------------------
// one workgroup per member: reduce the per-wave partials in a fixed order, decide, advance the clock
__global__ __launch_bounds__(kThreads) void k_control(DevCtx c, int n_partials, double two_n_free, int n_timepoints) {
  const int m = blockIdx.x;
  __shared__ double red[kThreads];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n_partials; i += kThreads) acc += c.err_partial[(size_t)m * c.n_wg * 4 + i];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = kThreads / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x != 0) return;
  Clock ck = c.clock[m];
  if (ck.state) return;
  if (ck.fin_next) { ck.state = 1; ck.accept = 0; c.clock[m] = ck; return; }
  const double ratio = sqrt(red[0] / two_n_free);
  ck.attempts++;
  if (!(ratio == ratio)) { ck.state = 2; c.clock[m] = ck; return; }
  const double h_new = dopri_next_step(ck.h, ratio);
  ck.accept = ratio <= 1.0;
  if (ck.accept) {
    ck.t_last = ck.t; ck.h_acc = ck.h; ck.t = ck.t + ck.h; ck.accepted++;
    ck.out_lo = ck.out_idx;
    if (c.acc_times && ck.accepted <= c.acc_cap) c.acc_times[(size_t)m * c.acc_cap + ck.accepted - 1] = ck.t;
    if (c.step_counts && n_timepoints > 1) c.step_counts[(size_t)m * (n_timepoints - 1) + min(max(ck.out_idx - 1, 0), n_timepoints - 2)]++;
    while (ck.out_idx < n_timepoints && c.ts_dev[ck.out_idx] <= ck.t) ck.out_idx++;
    ck.out_hi = ck.out_idx;
    if (ck.out_idx >= n_timepoints) ck.fin_next = 1;
  }
  ck.h = h_new;
  if (!(h_new > 0.0)) ck.state = 3;
  c.clock[m] = ck;
}

// elementwise: dense output for the outputs crossed by an accepted step, commit (y_n <- y1, k_1 <- k_7), and the
// stage-1 record of the next attempt with the new step size.  cm / cma: mid-point weights (velocity / position form).
struct DenseCoef { double cm[7], cma[7]; double a10; };

__global__ __launch_bounds__(kThreads) void k_prepare(DevCtx c, DenseCoef dc, int n_timepoints) {
  const int m = blockIdx.y + c.m0;
  const int slot = logical_wg(blockIdx.x, c.n_wg) * kThreads + threadIdx.x;
  if (slot >= c.n_slots) return;
  const Clock ck = c.clock[m];
  if (ck.state) return;
  const int b = slot >> 2, k = slot & 3, kd = k < 3 ? k : 2;
  const size_t nd = (size_t)c.n_blocks * 3;
  const int dof = b * 3 + kd;
  double* POS0 = c.POS + ((size_t)m * c.nbuf + 0) * c.n_blocks * kPos + (size_t)b * kPos;
  double* VEL0 = c.VEL + ((size_t)m * c.nbuf + 0) * nd;
  const double* POS3 = c.POS + ((size_t)m * c.nbuf + 3) * c.n_blocks * kPos + (size_t)b * kPos;
  const double* VEL3 = c.VEL + ((size_t)m * c.nbuf + 3) * nd;
  double* Am = c.A + (size_t)m * (c.s + 1) * nd;
  double qn = POS0[kd], vn = VEL0[dof], a0 = Am[dof];
  const int sidx = c.block_special[b];
  const bool constrained = sidx >= 0 && k < 3 && ((c.special[sidx].con_mask >> k) & 1);
  if (ck.accept) {
    const double q1 = POS3[kd], v1 = VEL3[dof], a6 = Am[(size_t)6 * nd + dof];
    if (k < 3 && ck.out_hi > ck.out_lo) {
      const double h = ck.h_acc;
      double sm = dc.cm[0] * a0 + dc.cm[6] * a6, sma = dc.cma[0] * a0 + dc.cma[6] * a6;
      for (int l = 1; l < 6; ++l) { const double al = Am[(size_t)l * nd + dof]; sm += dc.cm[l] * al; sma += dc.cma[l] * al; }
      const double qmid = qn + h * (0.5 * vn + h * sma), vmid = vn + h * sm;
      for (int kk = ck.out_lo; kk < ck.out_hi; ++kk) {
        const double tk = c.ts_dev[kk];
        const double r = (tk - ck.t_last) / (ck.t - ck.t_last);
        double oq = dopri_dense(qn, q1, qmid, vn, v1, h, r), ov = dopri_dense(vn, v1, vmid, a0, a6, h, r);
        if (constrained) { TimeVals tv = constrained_value(c, m, c.special[sidx], k, tk); oq = tv.g; ov = tv.gt; }
        double* f = c.fields_dev + ((size_t)m * n_timepoints + kk) * c.n_blocks * 6;
        f[dof] = oq;
        f[nd + dof] = ov;
      }
    }
    // commit
    if (k < 3) {
      *reinterpret_cast<double2*>(POS0 + 2 * k) = *reinterpret_cast<const double2*>(POS3 + 2 * k);
      VEL0[dof] = v1;
      Am[dof] = a6;
    }
    qn = q1; vn = v1; a0 = a6;
  }
  // stage-1 record of the next attempt: Q_1 = q_n + h a10 v_n, V_1 = v_n + h a10 A_0
  double qnext = qn + ck.h * dc.a10 * vn, vnext = vn + ck.h * dc.a10 * a0;
  if (constrained) { TimeVals tv = constrained_value(c, m, c.special[sidx], k, ck.t + dc.a10 * ck.h); qnext = tv.g; vnext = tv.gt; }
  const double y1 = quad_bcast<1>(qnext), th2 = quad_bcast<2>(qnext);
  double sn, cs;
  fast_sincos(0.5 * th2, &sn, &cs);
  const double2 chunk = k == 0 ? make_double2(qnext, y1) : (k == 1 ? make_double2(th2, cs) : make_double2(sn, 0.0));
  if (k < 3) {
    *reinterpret_cast<double2*>(c.POS + ((size_t)m * c.nbuf + 1) * c.n_blocks * kPos + (size_t)b * kPos + 2 * k) = chunk;
    c.VEL[((size_t)m * c.nbuf + 1) * nd + dof] = vnext;
  }
}

// per-member time for k_init (initial-step probe): records of state `y` at time tm[m] into buffer buf
__global__ __launch_bounds__(kThreads) void k_init_tm(DevCtx c, const double* state0, const double* tm, int buf) {
  const int m = blockIdx.y + c.m0;
  const int tid = blockIdx.x * kThreads + threadIdx.x;
  if (tid >= c.n_slots) return;
  const int b = tid >> 2, d = tid & 3;
  if (d == 3) return;
  const size_t nd = (size_t)c.n_blocks * 3;
  double q = state0[(size_t)m * 2 * nd + b * 3 + d], v = state0[(size_t)m * 2 * nd + nd + b * 3 + d];
  const int sidx = c.block_special[b];
  if (sidx >= 0 && ((c.special[sidx].con_mask >> d) & 1)) {
    TimeVals tv = constrained_value(c, m, c.special[sidx], d, tm[m]);
    q = tv.g; v = tv.gt;
  }
  double* pr = c.POS + ((size_t)m * c.nbuf + buf) * c.n_blocks * kPos + (size_t)b * kPos;
  pr[d] = q;
  c.VEL[((size_t)m * c.nbuf + buf) * nd + b * 3 + d] = v;
  if (d == 2) {
    double sn, cs;
    fast_sincos(0.5 * q, &sn, &cs);
    pr[3] = cs; pr[4] = sn; pr[5] = 0.0;
  }
}

template <int MODEL, int CONTACT>
__global__ __launch_bounds__(kThreads) void k_energy(DevCtx c, double* e_slot) {
  const int m = blockIdx.y + c.m0;
  const int slot = blockIdx.x * kThreads + threadIdx.x;
  if (slot >= c.n_slots) return;
  LaneIn L;
  load_lane<CONTACT>(c, m, slot, pos_in(c, m, 0, 0), L);
  double e = 0.0;
  if (L.info >= 0 && !(L.info & 1)) {
    BondGrad<double> g;
    bond_grad<MODEL, double>(L.o, L.p, L.rox, L.roy, L.rpx, L.rpy, L.lx, L.ly, L.l0, L.il0, L.ks, L.ksh, L.kr, L.sgn, g);
    e = g.e;
    if (CONTACT) {
      ContactGrad<double> cg;
      contact_grad<double>(L.sgn * (L.o.th - L.p.th), L.phi1, L.phi2, L.am, L.ac, L.kc, cg);
      e += cg.e;
    }
  }
  e_slot[(size_t)m * c.n_slots + slot] = e;
}

// ---- stage checkpoint: rebuild a stage record in the reverse sweep --------------------------------------------------
// Record r (1 <= r < s) of step nr from what the forward pass kept: (q, v) of the step in the trajectory checkpoint and
// the stage accelerations A_0 .. A_{r-1} of that step in AD.  The stage state of a DOF depends on its own history only,
// so this is elementwise; it is the forward epilogue run again.  All four lanes of a quad must call it (DPP).
//   rc: stage_coef(tableau, r - 1)  (row r of the tableau; c_next = c_r),  h, t: size and start time of step nr
__device__ __forceinline__ void rebuild_record(const DevCtx& c, int m, int b, int k, const StageCoef& rc, int r, long long nr,
                                               double h, double t) {
  const int kd = k < 3 ? k : 2;
  const u32 nd = (u32)c.n_blocks * 3;
  const u32 o_dof = ((u32)b * 3 + kd) * 8;
  const double* tr = c.traj + (size_t)m * c.traj_stride + (size_t)nr * c.n_blocks * kStep;
  const double qn = ldg<double>(tr, ((u32)b * kPos + kd) * 8);
  const double vn = ldg<double>(tr + (size_t)c.n_blocks * kPos, o_dof);
  const int sidx = ldg<int>(c.block_special, (u32)b * 4);
  const double* Ad = c.AD + (size_t)m * c.ad_stride + (size_t)nr * ((u32)(c.s - 1) * nd);
  double al[kMaxStages - 1];
#pragma unroll
  for (int l = 0; l < kMaxStages - 1; ++l) al[l] = l < r ? ldg<double>(Ad + (size_t)l * nd, o_dof) : 0.0;
  double sv = 0.0, sq = 0.0;
#pragma unroll
  for (int l = 0; l < kMaxStages - 1; ++l) { sv += rc.cv[l] * al[l]; sq += rc.cq[l] * al[l]; }
  double qnext = qn + h * (rc.c_next * vn + h * sq);
  double vnext = vn + h * sv;
  if (k < 3 && sidx >= 0 && ((c.special[sidx].con_mask >> k) & 1)) {
    const TimeVals tv = constrained_value(c, m, c.special[sidx], k, t + rc.c_next * h);
    qnext = tv.g; vnext = tv.gt;
  }
  const double y1 = quad_bcast<1>(qnext), th2 = quad_bcast<2>(qnext);
  double sn, cs;
  fast_sincos(0.5 * th2, &sn, &cs);
  const double2 chunk = k == 0 ? make_double2(qnext, y1) : (k == 1 ? make_double2(th2, cs) : make_double2(sn, 0.0));
  if (k < 3) {
    stg<double2>(c.POS + ((size_t)m * c.nbuf + r) * (u32)c.n_blocks * kPos, ((u32)b * kPos + 2 * k) * 8, chunk);
    stg<double>(c.VEL + ((size_t)m * c.nbuf + r) * nd, o_dof, vnext);
  }
}

// record s-1 of the LAST step, before the reverse sweep starts (every later record is rebuilt by the reverse launch
// that precedes its reader)
__global__ __launch_bounds__(kThreads) void k_rebuild_first(DevCtx c, StageCoef rc, int r, long long nr, double h, double t) {
  const int m = blockIdx.y + c.m0;
  const int slot = logical_wg(blockIdx.x, c.n_wg) * kThreads + threadIdx.x;
  if (slot >= c.n_slots) return;
  rebuild_record(c, m, slot >> 2, slot & 3, rc, r, nr, h, t);
}

// ---- reverse stage ---------------------------------------------------------------------------
//   in_buf: stage buffer with the stage records (recomputed), or -1: the checkpoint of step n (i == 0)
//   wbuf_static: >= 0 selects the (w, kbar_q) input buffer (test hook); -1: parity of the stage ordinal
//   BOND_GRADS: also accumulate d/d(reference vector, stiffnesses, contact constants) (only when the caller asks for them:
//   a compile-time switch, the dual parts of those derivatives are dead code otherwise)
template <int MODEL, int CONTACT, int BOND_GRADS>
//   rb > 0 (stage checkpoint): after its own work the launch rebuilds stage record rb -- of the same step when i >= 2
//   (rb = i - 1, read by the next reverse launch), of the previous step when i == 0 (rb = s - 1); rc = stage_coef(rb - 1)
__global__ __launch_bounds__(kThreads) DFX_ADJ_OCC void k_adj_stage(DevCtx c, AdjCoef ac, int i, int j, int in_buf, int wbuf_static,
                                                        int local_only, StageCoef rc, int rb) {
  const int m = blockIdx.y + c.m0;
  const int slot = logical_wg(blockIdx.x, c.n_wg) * kThreads + threadIdx.x;
  if (slot >= c.n_slots) return;
  const int b = slot >> 2, k = slot & 3, kd = k < 3 ? k : 2;
  const Seg sg = *c.cur;
  const long long n = sg.base_step + j;
  // the reverse sweep visits forward ordinals n*s+i in decreasing order, so the buffer parity alternates
  const int win = wbuf_static >= 0 ? wbuf_static : (int)((n * c.s + i) & 1);
  const u32 nd = (u32)c.n_blocks * 3, nd6 = (u32)c.n_blocks * 6;
  // ---- load phase
  const double* POSin = pos_in(c, m, in_buf, n);
  const MemberBases B = member_bases(c, m);
  LaneRaw R;
  issue_lane<CONTACT>(c, B, slot, POSin, R);
  const int dof = b * 3 + kd;
  const u32 o_dof = (u32)dof * 8, o_b6 = ((u32)b * 6 + kd) * 8;          // per-lane byte offsets shared by the per-DOF arrays
  const double* Win = c.W + ((size_t)m * 2 + win) * nd;
  const double w_d = ldg<double>(Win, o_dof);
  // partner's w from the guessed slot (same batch as everything else)
  double wpx, wpy, wpth;
  { const u32 gb = (u32)(R.guess >> 2) * 24; wpx = ldg<double>(Win, gb); wpy = ldg<double>(Win, gb + 8); wpth = ldg<double>(Win, gb + 16); }
  const double v_i = ldg<double>(vel_in(c, m, in_buf, n), o_dof);
  const double kq_in = ldg<double>(c.KQ + ((size_t)m * 2 + win) * nd, o_dof);
  const double damp = c.damping_uniform ? B.cst[6 + kd] : ldg<double>(c.damping + (size_t)m * nd, o_dof);
  const double invm = ldg<double>(c.inv_m + (size_t)m * nd, o_dof);
  const int sidx = ldg<int>(c.block_special, (u32)b * 4);
  double* YBm = c.YB + (size_t)m * (u32)c.s * nd6;
  double* LAMm = c.LAM + (size_t)m * nd6;
  double lq = 0.0, lv = 0.0, sq = 0.0, sv = 0.0;
  if (!local_only) {
    lq = ldg<double>(LAMm, o_b6); lv = ldg<double>(LAMm, o_b6 + 24);
    double yq[kMaxStages], yv[kMaxStages];
#pragma unroll
    for (int jj = 1; jj < kMaxStages; ++jj) {       // all loads issued together
      const bool on = jj > i && jj < c.s;
      yq[jj] = on ? ldg<double>(YBm + (size_t)jj * nd6, o_b6) : 0.0;
      yv[jj] = on ? ldg<double>(YBm + (size_t)jj * nd6, o_b6 + 24) : 0.0;
    }
#pragma unroll
    for (int jj = 1; jj < kMaxStages; ++jj) {
      const double cf = i > 0 ? ac.col[jj] : 1.0;
      sq += cf * yq[jj];
      sv += cf * yv[jj];
    }
  }
  LaneIn L;
  resolve_lane<CONTACT>(c, B, POSin, R, L);
  if (L.pslot != L.guess) { const u32 pb = (u32)(L.pslot >> 2) * 24; wpx = ldg<double>(Win, pb); wpy = ldg<double>(Win, pb + 8); wpth = ldg<double>(Win, pb + 16); }
  const double wox = quad_bcast<0>(w_d), woy = quad_bcast<1>(w_d), woth = quad_bcast<2>(w_d);
  // ---- Hessian-vector product + mixed parameter derivatives of this slot
  double hx = 0.0, hy = 0.0, hth = 0.0;
  double ex = 0.0, ey = 0.0, eth = 0.0;   // dE/du of this slot (value parts): gives the stage acceleration without re-reading it
  if (L.info >= 0) {
    BlockRec<Dual> o = seed_rec(L.o, wox, woy, woth);
    BlockRec<Dual> p = seed_rec(L.p, wpx, wpy, wpth);
    BondGrad<Dual> g;
    bond_grad<MODEL, Dual>(o, p, L.rox, L.roy, L.rpx, L.rpy, L.lx, L.ly, L.l0, L.il0, L.ks, L.ksh, L.kr, L.sgn, g);
    hx = g.fx.e; hy = g.fy.e; hth = g.fth.e;
    ex = g.fx.v; ey = g.fy.v; eth = g.fth.v;
    ContactGrad<Dual> cg;
    if (CONTACT) {
      contact_grad<Dual>(L.sgn * (o.th - p.th), L.phi1, L.phi2, L.am, L.ac, L.kc, cg);
      hth += L.sgn * cg.dkap.e;
      eth += L.sgn * cg.dkap.v;
    }
    // L += w . F = -w . grad E   =>   dL/dp = -eps(dE/dp)
    // Every accumulator address has exactly one writer per launch: plain load-add-store.  (Fire-and-forget L2 atomics,
    // global_atomic_add_f64 without return, would spare the round trip for the old value but were measured 10-25 %
    // slower per launch: four fp64 atomics per lane saturate the L2 atomic units.)
    const size_t ms = (size_t)m * (u32)c.n_slots;
    double* grm = c.g_r + ms * 2;
    double2 r = ldg<double2>(grm, (u32)slot * 16);
    r.x -= g.rx.e; r.y -= g.ry.e;
    stg<double2>(grm, (u32)slot * 16, r);
    // both ends hold the same contact dual: each accumulates one of the two void-angle derivatives (8 B per lane)
    if (CONTACT) stg<double>(c.g_phi + ms, (u32)slot * 8, ldg<double>(c.g_phi + ms, (u32)slot * 8) - ((L.info & 1) ? cg.p2.e : cg.p1.e));
    if (!(L.info & 1)) {
      if (BOND_GRADS) {
        double* q = c.g_b + (ms + slot) * 8;
        q[0] -= g.lx.e; q[1] -= g.ly.e; q[2] -= g.ks.e; q[3] -= g.ksh.e; q[4] -= g.kr.e;
        if (CONTACT) { q[5] -= cg.am.e; q[6] -= cg.ac.e; q[7] -= cg.kc.e; }
      }
    }
  }
  hx = quad_sum(hx);
  hy = quad_sum(hy);
  hth = quad_sum(hth);
  ex = quad_sum(ex);
  ey = quad_sum(ey);
  eth = quad_sum(eth);
  // ---- DOF epilogue
  double h = sg.h, t_n = sg.t_interval + (sg.j0 + j) * sg.h, h_before = (sg.j0 + j) == 0 ? sg.h_prev : sg.h;
  if (c.t_steps) { t_n = c.t_steps[n]; h = c.t_steps[n + 1] - t_n; h_before = n > 0 ? t_n - c.t_steps[n - 1] : 0.0; }
  if (k < 3) {
    const double hw = k == 0 ? hx : (k == 1 ? hy : hth);
    const double dE = k == 0 ? ex : (k == 1 ? ey : eth);
    bool constrained = false;
    double fload = 0.0;
    if (sidx >= 0) {
      const dfx_special& sp = c.special[sidx];
      constrained = (sp.con_mask >> k) & 1;
      const double t_i = t_n + ac.c_i * h;
      double gp[kMaxFnParams];
      for (int f = 0; f < c.n_fns; ++f) {
        const double coef = constrained ? -hw * sp.con_coef[k][f] : w_d * sp.load_coef[k][f];
        const bool loaded = !constrained && sp.load_coef[k][f] != 0.0;
        if ((coef != 0.0 && c.fn_g) || loaded) {
          double g, gt;
          eval_time_fn(c.fns[(u32)m * DFX_MAX_FNS + f], t_i, g, gt, gp);
          if (loaded) fload += sp.load_coef[k][f] * g;
          if (coef != 0.0 && c.fn_g) {
            double* q = c.fn_g + (((size_t)m * c.n_special + sidx) * DFX_MAX_FNS + f) * DFX_FN_PARAMS;
            for (int kk = 0; kk < DFX_FN_PARAMS; ++kk) acc_add(q + kk, coef * gp[kk]);   // up to 3 DOF lanes of a block share q
          }
        }
      }
    }
    const double a_i = constrained ? 0.0 : (fload - dE - damp * v_i) * invm;
    double ybq = 0.0, ybv = 0.0;
    if (!constrained) {
      ybq = -hw;
      ybv = kq_in - damp * w_d;
      double* bm = c.blk_m + (size_t)m * nd;
      stg<double>(bm, o_dof, ldg<double>(bm, o_dof) - w_d * a_i);
      if (c.blk_c) { double* bc = c.blk_c + (size_t)m * nd; stg<double>(bc, o_dof, ldg<double>(bc, o_dof) - w_d * v_i); }
    }
    stg<double>(YBm + (size_t)i * nd6, o_b6, ybq);
    stg<double>(YBm + (size_t)i * nd6, o_b6 + 24, ybv);
    if (!local_only) {
      double kq, kv;
      if (i > 0) {
        kq = h * (ac.col[c.s] * lq + ac.col[i] * ybq + sq);
        kv = h * (ac.col[c.s] * lv + ac.col[i] * ybv + sv);
      } else {
        lq += ybq + sq;
        lv += ybv + sv;
        const bool first = (sg.j0 + j) == 0;
        if (first && c.G && !constrained) {
          const double* G = c.G + ((size_t)sg.interval * c.batch + m) * (size_t)nd6;
          lq += G[b * 6 + k]; lv += G[b * 6 + 3 + k];
        }
        if (constrained) { lq = 0.0; lv = 0.0; }
        stg<double>(LAMm, o_b6, lq);
        stg<double>(LAMm, o_b6 + 24, lv);
        kq = h_before * ac.col[c.s] * lq;
        kv = h_before * ac.col[c.s] * lv;
      }
      stg<double>(c.KQ + ((size_t)m * 2 + (win ^ 1)) * nd, o_dof, kq);
      stg<double>(c.W + ((size_t)m * 2 + (win ^ 1)) * nd, o_dof, constrained ? 0.0 : kv * invm);
    }
  }
  if (rb > 0) {
    if (i > 0) rebuild_record(c, m, b, k, rc, rb, n, h, t_n);
    else if (n > 0) rebuild_record(c, m, b, k, rc, rb, n - 1, h_before, c.t_steps ? c.t_steps[n - 1] : t_n - h_before);
  }
}

// start of the reverse sweep: lambda_N = G_last; kbar_{s-1} of the last step into buffer `buf`
__global__ __launch_bounds__(kThreads) void k_adj_begin(DevCtx c, double h_last, double b_last, int buf) {
  const int m = blockIdx.y + c.m0;
  const int tid = blockIdx.x * kThreads + threadIdx.x;
  if (tid >= c.n_slots) return;
  const int b = tid >> 2, d = tid & 3;
  if (d == 3) return;
  const size_t nd = (size_t)c.n_blocks * 3, nd6 = (size_t)c.n_blocks * 6;
  const int sidx = c.block_special[b];
  const bool con = sidx >= 0 && ((c.special[sidx].con_mask >> d) & 1);
  const double* G = c.G + ((size_t)(c.n_timepoints - 1) * c.batch + m) * nd6;
  const double lq = con ? 0.0 : G[b * 6 + d], lv = con ? 0.0 : G[b * 6 + 3 + d];
  c.LAM[(size_t)m * nd6 + b * 6 + d] = lq;
  c.LAM[(size_t)m * nd6 + b * 6 + 3 + d] = lv;
  c.KQ[((size_t)m * 2 + buf) * nd + b * 3 + d] = h_last * b_last * lq;
  c.W[((size_t)m * 2 + buf) * nd + b * 3 + d] = con ? 0.0 : h_last * b_last * lv * c.inv_m[(size_t)m * nd + b * 3 + d];
}

// test hook: w = lam_v / m, kbar_q = lam_q  into buffer 0
__global__ __launch_bounds__(kThreads) void k_seed_vjp(DevCtx c, const double* lam) {
  const int m = blockIdx.y + c.m0;
  const int tid = blockIdx.x * kThreads + threadIdx.x;
  if (tid >= c.n_blocks * 3) return;
  const int b = tid / 3, d = tid % 3;
  const size_t nd = (size_t)c.n_blocks * 3;
  const int sidx = c.block_special[b];
  const bool con = sidx >= 0 && ((c.special[sidx].con_mask >> d) & 1);
  const double* l = lam + (size_t)m * c.n_blocks * 6;
  c.W[(size_t)m * 2 * nd + tid] = con ? 0.0 : l[nd + tid] * c.inv_m[(size_t)m * nd + tid];
  c.KQ[(size_t)m * 2 * nd + tid] = l[tid];
}

// fields_bar (batch, T, 2, n_blocks, 3) -> G (T, batch, n_blocks, 6)
__global__ __launch_bounds__(kThreads) void k_pack_G(DevCtx c, const double* fields_bar, double* G) {
  const size_t total = (size_t)c.batch * c.n_timepoints * c.n_blocks * 3;
  const size_t tid = (size_t)blockIdx.x * kThreads + threadIdx.x;
  if (tid >= total) return;
  const size_t mk = tid / ((size_t)c.n_blocks * 3);
  const int r = (int)(tid % ((size_t)c.n_blocks * 3));
  const int b = r / 3, d = r % 3;
  const double* f = fields_bar + mk * c.n_blocks * 6;
  const size_t m_ = mk / c.n_timepoints, k_ = mk % c.n_timepoints;
  double* g = G + (k_ * c.batch + m_) * c.n_blocks * 6;
  g[b * 6 + d] = f[r];
  g[b * 6 + 3 + d] = f[(size_t)c.n_blocks * 3 + r];
}

// kinetic-energy objective: G <- m v on target blocks; per-member objective by one workgroup
__global__ __launch_bounds__(kThreads) void k_kinetic(DevCtx c, const double* fields, const int32_t* target, int n_target,
                                                      double* G, double* objective) {
  const int m = blockIdx.x;
  __shared__ double red[kThreads];
  double acc = 0.0;
  const int per_t = n_target * 3;
  const long long total = (long long)c.n_timepoints * per_t;
  for (long long idx = threadIdx.x; idx < total; idx += kThreads) {
    const int k = (int)(idx / per_t), r = (int)(idx % per_t);
    const int b = target[r / 3], d = r % 3;
    const double v = fields[((size_t)m * c.n_timepoints + k) * c.n_blocks * 6 + (size_t)c.n_blocks * 3 + b * 3 + d];
    const double mass = 1.0 / c.inv_m[(size_t)m * c.n_blocks * 3 + b * 3 + d];
    acc += 0.5 * mass * v * v;
    if (G) G[((size_t)k * c.batch + m) * c.n_blocks * 6 + b * 6 + 3 + d] = mass * v;
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = kThreads / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0 && objective) objective[m] = red[0];
}

// explicit d(objective)/d(inertia) = sum_t v^2/2 on target DOFs, added to blk_m
__global__ void k_kinetic_mass_grad(DevCtx c, const double* fields, const int32_t* target, int n_target) {
  const int m = blockIdx.y + c.m0;
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_target * 3) return;
  const int b = target[r / 3], d = r % 3;
  double acc = 0.0;
  for (int k = 0; k < c.n_timepoints; ++k) {
    const double v = fields[((size_t)m * c.n_timepoints + k) * c.n_blocks * 6 + (size_t)c.n_blocks * 3 + b * 3 + d];
    acc += 0.5 * v * v;
  }
  c.blk_m[((size_t)m * c.n_blocks + b) * 3 + d] += acc;
}

}  // namespace

// ================================================================================================
// host side
// ================================================================================================
#define HIP_OK(call)                                                                                   \
  do {                                                                                                 \
    hipError_t e_ = (call);                                                                            \
    if (e_ != hipSuccess) {                                                                            \
      h->err = std::string(#call) + ": " + hipGetErrorString(e_);                                      \
      return 2;                                                                                        \
    }                                                                                                  \
  } while (0)

static std::string g_create_error;

template <class T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  hipError_t ensure(size_t count) {
    if (count <= n && p) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr; n = 0;
    hipError_t e = hipMalloc((void**)&p, std::max<size_t>(count, 1) * sizeof(T));
    if (e == hipSuccess) n = std::max<size_t>(count, 1);
    return e;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
};

// pinned host staging area: pageable hipMemcpy runs at < 1 GB/s on this platform, pinned DMA at PCIe rate
struct PinnedBuf {
  char* p = nullptr;
  size_t n = 0;
  hipError_t ensure(size_t bytes) {
    if (bytes <= n && p) return hipSuccess;
    if (p) (void)hipHostFree(p);
    p = nullptr; n = 0;
    hipError_t e = hipHostMalloc((void**)&p, std::max<size_t>(bytes, 64), hipHostMallocDefault);
    if (e == hipSuccess) n = std::max<size_t>(bytes, 64);
    return e;
  }
  void release() { if (p) (void)hipHostFree(p); p = nullptr; n = 0; }
};

struct Group {            // members [m0, m0+nm) advance on their own stream so launch bubbles of one group overlap work of another
  hipStream_t stream = nullptr;
  hipStream_t stream2 = nullptr;   // reverse sweep: recompute of step n-1 runs here while `stream` does the reverse stages of step n
  hipEvent_t done = nullptr;
  std::vector<hipEvent_t> ev_a, ev_b;  // per step of a segment: records ready / reverse stages done
  int m0 = 0, nm = 0;
};

struct dfx_handle {
  Plan pl;
  std::vector<Group> groups;
  bool dual_chain = true;
  hipEvent_t ev_fork2 = nullptr;
  PinnedBuf stage;
  hipEvent_t ev_fork = nullptr;
  PackedParams pp;
  std::string err;
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool have_params = false, have_traj = false, have_fields = false;
  bool use_graph = true;
  bool want_bond_grads = true, want_fn_grads = true, want_damping_grads = true;
  DevBuf<int32_t> d_slot_info, d_block_special;
  DevBuf<dfx_special> d_special;
  DevBuf<double> d_p_r, d_p_l, d_p_k, d_p_phi, d_cst, d_inv_m, d_damping, d_l_dict;
  DevBuf<uint8_t> d_l_idx;
  DevBuf<TimeFn> d_fns;
  DevBuf<double> d_fn_table[DFX_MAX_FNS];
  DevBuf<Seg> d_segs, d_cur;
  DevBuf<Clock> d_clock;
  DevBuf<double> d_err_partial, d_ts;
  double rtol = 0.0, atol = 0.0;
  bool adaptive = false;
  DevBuf<int> d_seg_idx;
  std::vector<Seg> segs;
  DevBuf<double> d_traj, d_POS, d_VEL, d_A, d_state0, d_fields;
  DevBuf<double> d_YB, d_LAM, d_W, d_KQ, d_G, d_g_r, d_g_phi, d_g_b, d_blk_m, d_blk_c, d_fn_g, d_tmp, d_obj;
  DevBuf<int32_t> d_target;
  std::vector<double> ts;
  std::vector<int> spis;           // RK steps in each output interval (fixed grid)
  std::vector<long long> step0;    // first step ordinal of each interval
  DevBuf<int> d_step_counts;
  int n_counts = 0;
  DevBuf<double> d_acc_times, d_tsteps;
  DevBuf<double> d_AD;             // stage checkpoint (stage accelerations of every step)
  bool dense = false;              // the last fixed-grid forward kept the stage checkpoint
  std::vector<double> t_steps;     // caller-chosen step boundaries (empty: equal steps)
  std::vector<long long> accepted_per_member;
  bool have_adaptive_record = false;
  long long n_total = 0;
  std::map<std::pair<int, int>, hipGraphExec_t> graphs;
  DevCtx graph_ctx_snapshot;
  bool graph_ctx_valid = false;
  long long launches = 0;
};

static void drop_graphs(dfx_handle* h) {
  for (auto& kv : h->graphs) (void)hipGraphExecDestroy(kv.second);
  h->graphs.clear();
  h->graph_ctx_valid = false;
}

static DevCtx make_ctx(dfx_handle* h) {
  const Plan& pl = h->pl;
  DevCtx c;
  memset(&c, 0, sizeof(c));
  c.n_blocks = pl.n_blocks; c.n_slots = pl.n_slots; c.n_fns = pl.n_fns; c.batch = pl.batch; c.s = pl.tab.s;
  { const char* a = getenv("DFX_ABLATE"); c.ablate = a ? atoi(a) : 0; }
  c.n_wg = (pl.n_slots + kThreads - 1) / kThreads;
  for (int k = 0; k < 4; ++k) c.pred[k] = pl.pred_delta[k];
  c.nbuf = 2 * pl.tab.s;
  c.n_special = pl.n_special; c.k_uniform = h->pp.k_uniform ? 1 : 0; c.n_timepoints = (int)h->ts.size();
  c.traj_stride = pl.batch ? (long long)(h->d_traj.n / pl.batch) : 0;
  c.slot_info = h->d_slot_info.p; c.block_special = h->d_block_special.p; c.special = h->d_special.p;
  c.p_lidx = h->d_l_idx.p; c.l_dict = h->d_l_dict.p; c.l_dict_on = h->pp.l_dict_ok ? 1 : 0; c.damping_uniform = h->pp.damping_uniform ? 1 : 0;
  c.p_r = h->d_p_r.p; c.p_l = h->d_p_l.p; c.p_k = h->d_p_k.p; c.p_phi = h->d_p_phi.p; c.cst = h->d_cst.p;
  c.inv_m = h->d_inv_m.p; c.damping = h->d_damping.p; c.fns = h->d_fns.p;
  c.cur = h->d_cur.p;
  c.clock = h->adaptive ? h->d_clock.p : nullptr;
  c.err_partial = h->d_err_partial.p; c.ts_dev = h->d_ts.p; c.fields_dev = h->d_fields.p;
  c.step_counts = h->adaptive ? h->d_step_counts.p : nullptr;
  c.acc_times = h->adaptive ? h->d_acc_times.p : nullptr; c.acc_cap = kAccCap;
  c.t_steps = (!h->adaptive && !h->t_steps.empty()) ? h->d_tsteps.p : nullptr;
  c.rtol = h->rtol; c.atol = h->atol;
  c.traj = h->have_traj ? h->d_traj.p : nullptr;
  c.AD = (h->have_traj && h->dense) ? h->d_AD.p : nullptr;
  c.ad_stride = pl.batch ? (long long)(h->d_AD.n / pl.batch) : 0;
  c.POS = h->d_POS.p; c.VEL = h->d_VEL.p; c.A = h->d_A.p;
  c.YB = h->d_YB.p; c.LAM = h->d_LAM.p; c.W = h->d_W.p; c.KQ = h->d_KQ.p; c.G = h->d_G.p;
  c.g_r = h->d_g_r.p; c.g_phi = h->d_g_phi.p; c.g_b = h->want_bond_grads ? h->d_g_b.p : nullptr;
  c.blk_m = h->d_blk_m.p; c.blk_c = h->want_damping_grads ? h->d_blk_c.p : nullptr;
  c.fn_g = h->want_fn_grads ? h->d_fn_g.p : nullptr;
  return c;
}

static dim3 slot_grid(const dfx_handle* h) { return dim3((h->pl.n_slots + kThreads - 1) / kThreads, h->pl.batch); }
static dim3 slot_grid(const dfx_handle* h, const Group& g) { return dim3((h->pl.n_slots + kThreads - 1) / kThreads, g.nm); }
// context of one group: same arrays, its own member range and its own segment cursor
static DevCtx group_ctx(const dfx_handle* h, const DevCtx& c, int gi) {
  DevCtx cg = c;
  cg.m0 = h->groups[gi].m0;
  cg.cur = h->d_cur.p + gi;
  return cg;
}

static StageCoef stage_coef(const Tableau& T, int i) {
  StageCoef sc;
  memset(&sc, 0, sizeof(sc));
  const int r = i + 1;
  for (int l = 0; l <= i; ++l) { sc.cv[l] = T.a[r][l]; sc.cq[l] = T.aa[r][l]; }
  sc.c_i = T.c[i];
  sc.c_next = T.c[r];
  return sc;
}
static AdjCoef adj_coef(const Tableau& T, int i) {
  AdjCoef ac;
  memset(&ac, 0, sizeof(ac));
  if (i > 0) {
    for (int j = i; j < T.s; ++j) ac.col[j] = T.a[j][i - 1];
    ac.col[T.s] = T.a[T.s][i - 1];
  } else {
    ac.col[T.s] = T.a[T.s][T.s - 1];
  }
  ac.c_i = T.c[i];
  return ac;
}

template <int MODEL, int CONTACT>
static void launch_fwd_t(dfx_handle* h, const DevCtx& c, hipStream_t st, dim3 grid, int i, int j, int in_buf, int out_buf, int y_buf, int mode) {
  hipLaunchKernelGGL((k_fwd_stage<MODEL, CONTACT>), grid, dim3(kThreads), 0, st, c, stage_coef(h->pl.tab, i), i, j, in_buf, out_buf, y_buf, mode);
}
static void launch_fwd(dfx_handle* h, const DevCtx& c, hipStream_t st, dim3 grid, int i, int j, int in_buf, int out_buf, int y_buf, int mode) {
  const Plan& pl = h->pl;
  if (pl.model == kNonlinear) { if (pl.contact) launch_fwd_t<kNonlinear, 1>(h, c, st, grid, i, j, in_buf, out_buf, y_buf, mode); else launch_fwd_t<kNonlinear, 0>(h, c, st, grid, i, j, in_buf, out_buf, y_buf, mode); }
  else { if (pl.contact) launch_fwd_t<kLinearized, 1>(h, c, st, grid, i, j, in_buf, out_buf, y_buf, mode); else launch_fwd_t<kLinearized, 0>(h, c, st, grid, i, j, in_buf, out_buf, y_buf, mode); }
  h->launches++;
}
static void launch_fwd(dfx_handle* h, const DevCtx& c, int i, int j, int in_buf, int out_buf, int y_buf, int mode) {
  launch_fwd(h, c, h->stream, slot_grid(h), i, j, in_buf, out_buf, y_buf, mode);
}
template <int MODEL, int CONTACT>
static void launch_adj_t(dfx_handle* h, const DevCtx& c, hipStream_t st, dim3 grid, int i, int j, int in_buf, int wbuf, int local_only) {
  // stage checkpoint: which record this launch rebuilds for the launch after it (0: none)
  const int s = h->pl.tab.s;
  const int rb = (c.AD && !local_only) ? (i >= 2 ? i - 1 : (i == 0 ? s - 1 : 0)) : 0;
  const StageCoef rc = stage_coef(h->pl.tab, rb > 0 ? rb - 1 : 0);
  if (c.g_b) hipLaunchKernelGGL((k_adj_stage<MODEL, CONTACT, 1>), grid, dim3(kThreads), 0, st, c, adj_coef(h->pl.tab, i), i, j, in_buf, wbuf, local_only, rc, rb);
  else hipLaunchKernelGGL((k_adj_stage<MODEL, CONTACT, 0>), grid, dim3(kThreads), 0, st, c, adj_coef(h->pl.tab, i), i, j, in_buf, wbuf, local_only, rc, rb);
}
static void launch_adj(dfx_handle* h, const DevCtx& c, hipStream_t st, dim3 grid, int i, int j, int in_buf, int wbuf, int local_only) {
  const Plan& pl = h->pl;
  if (pl.model == kNonlinear) { if (pl.contact) launch_adj_t<kNonlinear, 1>(h, c, st, grid, i, j, in_buf, wbuf, local_only); else launch_adj_t<kNonlinear, 0>(h, c, st, grid, i, j, in_buf, wbuf, local_only); }
  else { if (pl.contact) launch_adj_t<kLinearized, 1>(h, c, st, grid, i, j, in_buf, wbuf, local_only); else launch_adj_t<kLinearized, 0>(h, c, st, grid, i, j, in_buf, wbuf, local_only); }
  h->launches++;
}
static void launch_adj(dfx_handle* h, const DevCtx& c, int i, int j, int in_buf, int wbuf, int local_only) {
  launch_adj(h, c, h->stream, slot_grid(h), i, j, in_buf, wbuf, local_only);
}

// forward: stage i reads buffer fin(i), writes fout(i); buffer 0 is the step state
static int fin(int i) { return i == 0 ? 0 : 1 + ((i - 1) & 1); }
static int fout(int i, int s) { return i == s - 1 ? 0 : 1 + (i & 1); }

// enqueue one segment (kind 0: forward steps; kind 1: reverse steps) of group gi on that group's stream
static void enqueue_segment(dfx_handle* h, const DevCtx& cbase, int gi, int n_steps, int kind) {
  const int s = h->pl.tab.s;
  const Group& g = h->groups[gi];
  const DevCtx c = group_ctx(h, cbase, gi);
  const dim3 grid = slot_grid(h, g);
  hipLaunchKernelGGL(k_tick, dim3(1), dim3(1), 0, g.stream, (const Seg*)h->d_segs.p, h->d_seg_idx.p + 2 + gi, kind == 0 ? 1 : -1, h->d_cur.p + gi);
  h->launches++;
  if (kind == 0) {
    for (int j = 0; j < n_steps; ++j)
      for (int i = 0; i < s; ++i) launch_fwd(h, c, g.stream, grid, i, j, fin(i), fout(i, s), 0, (i == s - 1 && c.traj) ? 1 : 0);
  } else if (c.AD) {
    // stage checkpoint: no recompute launches; every reverse launch also rebuilds the record its successor reads
    for (int j = n_steps - 1; j >= 0; --j)
      for (int i = s - 1; i >= 0; --i) launch_adj(h, c, g.stream, grid, i, j, i == 0 ? -1 : i, -1, 0);
  } else if (!h->dual_chain) {
    for (int j = n_steps - 1; j >= 0; --j) {
      // recompute the stage records of step n from its checkpoint: stage i -> buffer i+1
      // (the acceleration of the last stage is re-derived inside its reverse launch, so s-1 recompute launches suffice)
      for (int i = 0; i < s - 1; ++i) launch_fwd(h, c, g.stream, grid, i, j, i == 0 ? -1 : i, i + 1, -1, 0);
      for (int i = s - 1; i >= 0; --i) launch_adj(h, c, g.stream, grid, i, j, i == 0 ? -1 : i, -1, 0);
    }
  } else {
    // Two chains: A(j) = recompute the stage records of step j (stream2), B(j) = its reverse stages (stream).
    // B(j) needs A(j); A(j-1) only needs the checkpoint, so it overlaps B(j).  Stage records are double-buffered by
    // step parity (set p: stage i in buffer 1 + p*(s-1) + i-1), hence A(j-1) must wait for B(j+1), the last reader of its set.
    Group& gm = h->groups[gi];
    while ((int)gm.ev_a.size() < n_steps) {
      hipEvent_t a = nullptr, b = nullptr;
      (void)hipEventCreateWithFlags(&a, hipEventDisableTiming);
      (void)hipEventCreateWithFlags(&b, hipEventDisableTiming);
      gm.ev_a.push_back(a); gm.ev_b.push_back(b);
    }
    auto buf = [&](int j, int i) { return 1 + (j & 1) * (s - 1) + (i - 1); };
    (void)hipEventRecord(h->ev_fork2, g.stream);                 // stream2 joins the capture / the sequence
    (void)hipStreamWaitEvent(g.stream2, h->ev_fork2, 0);
    for (int j = n_steps - 1; j >= 0; --j) {
      if (j + 2 <= n_steps - 1) (void)hipStreamWaitEvent(g.stream2, gm.ev_b[j + 2], 0);
      for (int i = 0; i < s - 1; ++i) launch_fwd(h, c, g.stream2, grid, i, j, i == 0 ? -1 : buf(j, i), buf(j, i + 1), -1, 0);
      (void)hipEventRecord(gm.ev_a[j], g.stream2);
      (void)hipStreamWaitEvent(g.stream, gm.ev_a[j], 0);
      for (int i = s - 1; i >= 0; --i) launch_adj(h, c, g.stream, grid, i, j, i == 0 ? -1 : buf(j, i), -1, 0);
      (void)hipEventRecord(gm.ev_b[j], g.stream);
    }
    // every stream2 operation is an ancestor of the last ev_a, which `stream` already waited for: the chains are joined
  }
}

static int run_segment(dfx_handle* h, const DevCtx& c, int gi, int n_steps, int kind) {
  if (!h->use_graph) { enqueue_segment(h, c, gi, n_steps, kind); return 0; }
  DevCtx key_ctx = c;
  key_ctx.n_timepoints = 0;  // not read by the stage kernels
  if (!h->graph_ctx_valid || memcmp(&h->graph_ctx_snapshot, &key_ctx, sizeof(DevCtx)) != 0) {
    drop_graphs(h);
    h->graph_ctx_snapshot = key_ctx;
    h->graph_ctx_valid = true;
  }
  auto key = std::make_pair(n_steps, kind * 64 + gi);
  auto it = h->graphs.find(key);
  const int s = h->pl.tab.s;
  const long long per = 1 + (long long)n_steps * ((kind == 0 || c.AD) ? s : 2 * s - 1);   // launches in the graph
  hipStream_t st = h->groups[gi].stream;
  if (it == h->graphs.end()) {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    const long long before = h->launches;
    HIP_OK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    enqueue_segment(h, c, gi, n_steps, kind);
    HIP_OK(hipStreamEndCapture(st, &graph));
    h->launches = before;
    HIP_OK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    (void)hipGraphDestroy(graph);
    it = h->graphs.emplace(key, exec).first;
  }
  HIP_OK(hipGraphLaunch(it->second, st));
  h->launches += per;
  return 0;
}

// the main stream has prepared the inputs: let every group stream start after it ...
static int fork_groups(dfx_handle* h) {
  HIP_OK(hipEventRecord(h->ev_fork, h->stream));
  for (auto& g : h->groups) if (g.stream != h->stream) HIP_OK(hipStreamWaitEvent(g.stream, h->ev_fork, 0));
  return 0;
}
// ... and the main stream continue after all of them
static int join_groups(dfx_handle* h) {
  for (auto& g : h->groups) {
    if (g.stream == h->stream) continue;
    HIP_OK(hipEventRecord(g.done, g.stream));
    HIP_OK(hipStreamWaitEvent(h->stream, g.done, 0));
  }
  return 0;
}

// One segment = one graph replay of n_steps steps inside one output interval.  Intervals with the most frequent step
// count are cut into chunks of kMaxGraphSteps; the others into power-of-two chunks, so that the number of distinct
// graphs stays <= log2(kMaxGraphSteps) + 3 whatever the counts are.
// Stage checkpoint: also keep the first s-1 stage accelerations of every step (+24 (s-1) B per unit and step on top of
// the 72 B of the state checkpoint) whenever that fits beside it: the reverse sweep then needs no recompute launches (s instead of
// 2s - 1 launches per step).  DFX_STAGE_CHECKPOINT=0/1 overrides the choice.  Returns whether the buffer is there.
static bool ensure_stage_checkpoint(dfx_handle* h, long long n_steps) {
  const Plan& pl = h->pl;
  const char* e = getenv("DFX_STAGE_CHECKPOINT");
  const size_t want = (size_t)pl.batch * (size_t)std::max<long long>(n_steps, 1) * (pl.tab.s - 1) * pl.n_blocks * 3;
  bool use = e ? e[0] != '0' : true;
  if (use && h->d_AD.n < want) {
    size_t free_b = 0, total_b = 0;
    // leave room: at most what is free minus 5 % of the device
    if (!e && (hipMemGetInfo(&free_b, &total_b) != hipSuccess || (want - h->d_AD.n) * sizeof(double) + total_b / 20 > free_b)) use = false;
    if (use && h->d_AD.ensure(want) != hipSuccess) { (void)hipGetLastError(); use = false; }
  }
  return use;
}

static void build_segments(dfx_handle* h) {
  h->segs.clear();
  const int Tn = (int)h->ts.size();
  // the most frequent count keeps whole-interval graphs (a run of K steps = many equal intervals + one shorter one)
  std::map<int, int> votes;
  for (int k = 0; k + 1 < Tn; ++k) ++votes[h->spis[k]];
  int common = 0, n_common = 0;
  for (auto& kv : votes) if (kv.second > n_common) { common = kv.first; n_common = kv.second; }
  for (int k = 0; k + 1 < Tn; ++k) {
    const int spi = h->spis[k];
    const double hh = (h->ts[k + 1] - h->ts[k]) / spi;
    const double hp = k > 0 ? (h->ts[k] - h->ts[k - 1]) / h->spis[k - 1] : 0.0;
    for (int j0 = 0; j0 < spi;) {
      int n = std::min(kMaxGraphSteps, spi - j0);
      if (spi != common) { int p2 = 1; while (p2 * 2 <= n) p2 *= 2; n = p2; }
      Seg sg;
      sg.t_interval = h->ts[k]; sg.h = hh; sg.h_prev = hp;
      sg.base_step = h->step0[k] + j0; sg.j0 = j0; sg.interval = k;
      sg.n_steps = n; sg.pad = 0;
      h->segs.push_back(sg);
      j0 += n;
    }
  }
}

static int ensure_work_buffers(dfx_handle* h) {
  const Plan& pl = h->pl;
  const size_t B = pl.batch, nb = pl.n_blocks, s = pl.tab.s;
  HIP_OK(h->d_POS.ensure(B * (2 * s) * nb * kPos));
  HIP_OK(h->d_VEL.ensure(B * (2 * s) * nb * 3));
  HIP_OK(h->d_A.ensure(B * (s + 1) * nb * 3));
  HIP_OK(h->d_state0.ensure(B * nb * 6));
  HIP_OK(h->d_cur.ensure(64));
  return 0;
}

static int ensure_adjoint_buffers(dfx_handle* h) {
  const Plan& pl = h->pl;
  const size_t B = pl.batch, nb = pl.n_blocks, s = pl.tab.s;
  const size_t nsp = std::max(1, pl.n_special);
  HIP_OK(h->d_YB.ensure(B * s * nb * 6));
  HIP_OK(h->d_LAM.ensure(B * nb * 6));
  HIP_OK(h->d_W.ensure(B * 2 * nb * 3));
  HIP_OK(h->d_KQ.ensure(B * 2 * nb * 3));
  HIP_OK(h->d_g_r.ensure(B * pl.n_slots * 2));
  HIP_OK(h->d_g_phi.ensure(B * pl.n_slots));
  HIP_OK(h->d_g_b.ensure(B * pl.n_slots * 8));
  HIP_OK(h->d_blk_m.ensure(B * nb * 3));
  HIP_OK(h->d_blk_c.ensure(B * nb * 3));
  HIP_OK(h->d_fn_g.ensure(B * nsp * DFX_MAX_FNS * DFX_FN_PARAMS));
  return 0;
}

static int zero_grad_accumulators(dfx_handle* h) {
  const Plan& pl = h->pl;
  const size_t B = pl.batch, nb = pl.n_blocks;
  const size_t nsp = std::max(1, pl.n_special);
  HIP_OK(hipMemsetAsync(h->d_g_r.p, 0, sizeof(double) * B * pl.n_slots * 2, h->stream));
  HIP_OK(hipMemsetAsync(h->d_g_phi.p, 0, sizeof(double) * B * pl.n_slots, h->stream));
  if (h->want_bond_grads) HIP_OK(hipMemsetAsync(h->d_g_b.p, 0, sizeof(double) * B * pl.n_slots * 8, h->stream));
  HIP_OK(hipMemsetAsync(h->d_blk_m.p, 0, sizeof(double) * B * nb * 3, h->stream));
  if (h->want_damping_grads) HIP_OK(hipMemsetAsync(h->d_blk_c.p, 0, sizeof(double) * B * nb * 3, h->stream));
  HIP_OK(hipMemsetAsync(h->d_fn_g.p, 0, sizeof(double) * B * nsp * DFX_MAX_FNS * DFX_FN_PARAMS, h->stream));
  return 0;
}

// download the accumulators and scatter them into dfx_grads
static int collect_grads(dfx_handle* h, dfx_grads* grads, bool with_state0) {
  // Only what the caller asked for crosses PCIe, and it is scattered straight from the pinned staging area into the
  // caller's arrays (a solve of a few steps is otherwise dominated by this function).
  const Plan& pl = h->pl;
  const size_t B = pl.batch, nb = pl.n_blocks, NS = pl.n_slots, nbd = pl.n_bonds;
  const size_t nsp = std::max(1, pl.n_special);
  HIP_OK(hipStreamSynchronize(h->stream));
  HIP_OK(hipGetLastError());
  if (!grads) return 0;
  const bool w_r = grads->centroid_node_vectors, w_phi = grads->void_angle0 && pl.contact;
  const bool w_b = h->want_bond_grads && (grads->reference_vector || grads->k_bond || grads->contact);
  const bool w_m = grads->inertia, w_c = grads->damping && h->want_damping_grads, w_fn = grads->fn_params && h->want_fn_grads;
  const bool w_lam = with_state0 && grads->state0;
  const size_t n_r = w_r ? B * NS * 2 : 0, n_phi = w_phi ? B * NS : 0, n_b = w_b ? B * NS * 8 : 0, n_m = w_m ? B * nb * 3 : 0,
               n_c = w_c ? B * nb * 3 : 0, n_fn = w_fn ? B * nsp * DFX_MAX_FNS * DFX_FN_PARAMS : 0, n_lam = w_lam ? B * nb * 6 : 0;
  HIP_OK(h->stage.ensure((n_r + n_phi + n_b + n_m + n_c + n_fn + n_lam + 8) * sizeof(double)));
  double* g_r = reinterpret_cast<double*>(h->stage.p);
  double* g_phi = g_r + n_r;
  double* g_b = g_phi + n_phi;
  double* g_m = g_b + n_b;
  double* g_c = g_m + n_m;
  double* fn_g = g_c + n_c;
  double* lam = fn_g + n_fn;
  auto pull = [&](double* dst, const double* src, size_t n) {
    return n ? hipMemcpyAsync(dst, src, sizeof(double) * n, hipMemcpyDeviceToHost, h->stream) : hipSuccess;
  };
  HIP_OK(pull(g_r, h->d_g_r.p, n_r));
  HIP_OK(pull(g_phi, h->d_g_phi.p, n_phi));
  HIP_OK(pull(g_b, h->d_g_b.p, n_b));
  HIP_OK(pull(g_m, h->d_blk_m.p, n_m));
  HIP_OK(pull(g_c, h->d_blk_c.p, n_c));
  HIP_OK(pull(fn_g, h->d_fn_g.p, n_fn));
  HIP_OK(pull(lam, h->d_LAM.p, n_lam));
  HIP_OK(hipStreamSynchronize(h->stream));
  HIP_OK(hipGetLastError());
  const int npb = pl.n_npb;
  if (w_r)
    for (size_t m = 0; m < B; ++m)
      for (size_t b = 0; b < nb; ++b)
        memcpy(grads->centroid_node_vectors + ((m * nb + b) * npb) * 2, g_r + (m * NS + b * kSlots) * 2, sizeof(double) * 2 * npb);
  if (grads->void_angle0) memset(grads->void_angle0, 0, sizeof(double) * B * nbd * 2);
  if (grads->reference_vector) memset(grads->reference_vector, 0, sizeof(double) * B * nbd * 2);
  if (grads->k_bond) memset(grads->k_bond, 0, sizeof(double) * B * nbd * 3);
  if (grads->contact) memset(grads->contact, 0, sizeof(double) * B * 3);
  if (w_phi || w_b)
    for (size_t m = 0; m < B; ++m)
      for (size_t sl = 0; sl < NS; ++sl) {
        const int info = pl.slot_info[sl];
        if (info < 0 || (info & 1)) continue;            // one entry per ligament: its end-0 slot
        const size_t bond = (size_t)pl.slot_bond[sl], i = m * NS + sl;
        if (w_phi) {
          grads->void_angle0[(m * nbd + bond) * 2] = g_phi[i];
          grads->void_angle0[(m * nbd + bond) * 2 + 1] = g_phi[m * NS + (size_t)(info >> 1)];
        }
        if (w_b) {
          const double* q = g_b + i * 8;
          if (grads->reference_vector) { grads->reference_vector[(m * nbd + bond) * 2] = q[0]; grads->reference_vector[(m * nbd + bond) * 2 + 1] = q[1]; }
          if (grads->k_bond) for (int c = 0; c < 3; ++c) grads->k_bond[(m * nbd + bond) * 3 + c] = q[2 + c];
          if (grads->contact) for (int c = 0; c < 3; ++c) grads->contact[m * 3 + c] += q[5 + c];
        }
      }
  if (grads->inertia) memcpy(grads->inertia, g_m, sizeof(double) * B * nb * 3);
  if (grads->damping) { if (w_c) memcpy(grads->damping, g_c, sizeof(double) * B * nb * 3); else memset(grads->damping, 0, sizeof(double) * B * nb * 3); }
  if (grads->fn_params) {
    const int W = DFX_MAX_FNS * DFX_FN_PARAMS;
    for (size_t m = 0; m < B; ++m)
      for (int f = 0; f < pl.n_fns; ++f)
        for (int i = 0; i < DFX_FN_PARAMS; ++i) {
          double acc = 0.0;
          if (w_fn) for (int sidx = 0; sidx < pl.n_special; ++sidx) acc += fn_g[(m * pl.n_special + sidx) * W + f * DFX_FN_PARAMS + i];
          grads->fn_params[(m * pl.n_fns + f) * DFX_FN_PARAMS + i] = acc;
        }
  }
  if (w_lam)
    for (size_t m = 0; m < B; ++m)
      for (size_t b = 0; b < nb; ++b)
        for (int d = 0; d < 3; ++d) {
          grads->state0[m * nb * 6 + b * 3 + d] = lam[m * nb * 6 + b * 6 + d];
          grads->state0[m * nb * 6 + nb * 3 + b * 3 + d] = lam[m * nb * 6 + b * 6 + 3 + d];
        }
  return 0;
}

static void set_grad_wishes(dfx_handle* h, const dfx_grads* g) {
  h->want_bond_grads = !g || g->reference_vector || g->k_bond || g->contact;
  h->want_fn_grads = !g || g->fn_params;
  h->want_damping_grads = !g || g->damping;
}

extern "C" {

int dfx_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char* dfx_version(void) { return "dfx-hip-gfx950 0.2.0"; }

const char* dfx_last_error(const dfx_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int dfx_create(const dfx_problem* problem, dfx_handle** out) {
  dfx_handle* h = new dfx_handle();
  auto fail = [&](int rc) { g_create_error = h->err; delete h; return rc; };
  if (build_plan(problem, h->pl, h->err)) return fail(1);
  {  // the stage kernels index per-handle arrays with 32 bits
    const Plan& pl = h->pl;
    const double B = pl.batch, nb = pl.n_blocks, st = pl.tab.s;
    const double largest = std::max({B * 2 * st * nb * kPos, B * pl.n_slots * 8.0, B * (st + 1) * nb * 6.0});
    if (largest >= 2147483648.0) {
      h->err = "create: batch x lattice too large for one handle (an array would exceed 2^31 elements); split the ensemble over several handles";
      return fail(1);
    }
  }
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0) { h->err = "no HIP device available (libdfx has no CPU fallback)"; return fail(2); }
  if (problem->device < 0 || problem->device >= ndev) { h->err = "device ordinal out of range"; return fail(1); }
  h->device = problem->device;
  if (hipSetDevice(h->device) != hipSuccess) { h->err = "hipSetDevice failed"; return fail(2); }
  if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) { h->err = "hipStreamCreate failed"; return fail(2); }
  for (int f = 0; f < DFX_MAX_FNS; ++f) {     // recorded input signals: static data, read by the few lanes that own driven DOFs
    if (h->pl.fn_table[f].empty()) continue;
    if (h->d_fn_table[f].ensure(h->pl.fn_table[f].size()) != hipSuccess ||
        hipMemcpy(h->d_fn_table[f].p, h->pl.fn_table[f].data(), sizeof(double) * h->pl.fn_table[f].size(), hipMemcpyHostToDevice) != hipSuccess) {
      h->err = "create: cannot upload the table of a time function"; return fail(2);
    }
    h->pl.fn_table_ptr[f] = h->d_fn_table[f].p;
  }
  (void)hipEventCreate(&h->ev0);
  (void)hipEventCreate(&h->ev1);
  const char* g = getenv("DFX_NO_GRAPH");
  h->use_graph = !(g && g[0] == '1');
  {
    // member groups on concurrent streams hide the launch boundary of one group behind the work of another, but only
    // when a group still fills the chip: measured best 2 groups at >= 2 waves per SIMD in total (128x128 x 4..16
    // members), 1 group below that (24x16 x 32 members: 4.8 s vs 8.4 s with 4 groups)
    const char* e = getenv("DFX_STREAMS");
    const long long waves = (long long)h->pl.batch * ((h->pl.n_slots + 63) / 64);
    int want = e ? atoi(e) : (waves >= 2048 ? 2 : 1);
    int ng = std::max(1, std::min(want, h->pl.batch));
    (void)hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming);
    (void)hipEventCreateWithFlags(&h->ev_fork2, hipEventDisableTiming);
    // the recompute/reverse overlap pays when the chip is otherwise idle (one system: measured -20 % reverse time for one
    // 128x128 system) and hurts once member groups already fill the 4 hardware queues (measured +50 % with 4 members)
    { const char* d = getenv("DFX_DUAL_CHAIN"); h->dual_chain = d ? d[0] != '0' : h->pl.batch == 1; }
    for (int gi = 0; gi < ng; ++gi) {
      Group gr;
      const int base = h->pl.batch / ng, rem = h->pl.batch % ng;
      gr.m0 = gi * base + std::min(gi, rem);
      gr.nm = base + (gi < rem ? 1 : 0);
      if (gi == 0) gr.stream = h->stream;   // group 0 rides on the main stream (HIP multiplexes streams onto few hardware queues)
      else if (hipStreamCreateWithFlags(&gr.stream, hipStreamNonBlocking) != hipSuccess) { h->err = "hipStreamCreate failed"; return fail(2); }
      (void)hipEventCreateWithFlags(&gr.done, hipEventDisableTiming);
      if (hipStreamCreateWithFlags(&gr.stream2, hipStreamNonBlocking) != hipSuccess) { h->err = "hipStreamCreate failed"; return fail(2); }
      h->groups.push_back(gr);
    }
  }
  const Plan& pl = h->pl;
  bool ok = h->d_slot_info.ensure(pl.n_slots) == hipSuccess && h->d_block_special.ensure(pl.n_blocks) == hipSuccess &&
            h->d_special.ensure(std::max(1, pl.n_special)) == hipSuccess && h->d_seg_idx.ensure(2 + 64) == hipSuccess &&
            h->d_cur.ensure(64) == hipSuccess;
  if (!ok) { h->err = "hipMalloc (static tables) failed"; return fail(2); }
  (void)hipMemcpy(h->d_slot_info.p, pl.slot_info.data(), sizeof(int32_t) * pl.n_slots, hipMemcpyHostToDevice);
  (void)hipMemcpy(h->d_block_special.p, pl.block_special.data(), sizeof(int32_t) * pl.n_blocks, hipMemcpyHostToDevice);
  if (pl.n_special)
    (void)hipMemcpy(h->d_special.p, pl.special.data(), sizeof(dfx_special) * pl.n_special, hipMemcpyHostToDevice);
  *out = h;
  return 0;
}

int dfx_destroy(dfx_handle* h) {
  if (!h) return 0;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  drop_graphs(h);
  h->d_slot_info.release(); h->d_block_special.release(); h->d_special.release();
  h->d_p_r.release(); h->d_p_l.release(); h->d_p_k.release(); h->d_p_phi.release(); h->d_cst.release(); h->d_l_dict.release(); h->d_l_idx.release();
  h->d_inv_m.release(); h->d_damping.release(); h->d_fns.release();
  for (int f = 0; f < DFX_MAX_FNS; ++f) h->d_fn_table[f].release();
  h->d_segs.release(); h->d_cur.release(); h->d_seg_idx.release(); h->d_clock.release(); h->d_err_partial.release(); h->d_ts.release(); h->d_step_counts.release(); h->d_acc_times.release(); h->d_tsteps.release(); h->d_AD.release();
  h->d_traj.release(); h->d_POS.release(); h->d_VEL.release(); h->d_A.release(); h->d_state0.release(); h->d_fields.release();
  h->d_YB.release(); h->d_LAM.release(); h->d_W.release(); h->d_KQ.release(); h->d_G.release();
  h->d_g_r.release(); h->d_g_phi.release(); h->d_g_b.release(); h->d_blk_m.release(); h->d_blk_c.release(); h->d_fn_g.release();
  h->d_tmp.release(); h->d_obj.release(); h->d_target.release(); h->stage.release();
  for (auto& gr : h->groups) { for (auto e : gr.ev_a) (void)hipEventDestroy(e); for (auto e : gr.ev_b) (void)hipEventDestroy(e); if (gr.stream2) (void)hipStreamDestroy(gr.stream2); if (gr.done) (void)hipEventDestroy(gr.done); if (gr.stream && gr.stream != h->stream) (void)hipStreamDestroy(gr.stream); }
  if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
  if (h->ev_fork2) (void)hipEventDestroy(h->ev_fork2);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return 0;
}

int dfx_set_params(dfx_handle* h, const dfx_params* params) {
  HIP_OK(hipSetDevice(h->device));
  const bool timing = getenv("DFX_TIMING") != nullptr;
  auto t0 = std::chrono::steady_clock::now();
  if (pack_params(h->pl, params, h->pp, h->err)) return 1;
  auto t1 = std::chrono::steady_clock::now();
  const PackedParams& pp = h->pp;
  {
    constexpr int NA = 8;
    const std::vector<double>* src[NA] = {&pp.p_r, pp.l_dict_ok ? nullptr : &pp.p_l, pp.k_uniform ? nullptr : &pp.p_k, &pp.p_phi, &pp.cst, &pp.inv_m,
                                          pp.damping_uniform ? nullptr : &pp.damping, pp.l_dict_ok ? &pp.l_dict : nullptr};
    DevBuf<double>* dst[NA] = {&h->d_p_r, &h->d_p_l, &h->d_p_k, &h->d_p_phi, &h->d_cst, &h->d_inv_m, &h->d_damping, &h->d_l_dict};
    size_t total = pp.l_idx.size();
    for (int i = 0; i < NA; ++i) if (src[i]) total += src[i]->size() * sizeof(double);
    HIP_OK(h->stage.ensure(total + 64));
    size_t off = 0;
    if (pp.l_dict_ok) {
      HIP_OK(h->d_l_idx.ensure(pp.l_idx.size()));
      memcpy(h->stage.p, pp.l_idx.data(), pp.l_idx.size());
      HIP_OK(hipMemcpyAsync(h->d_l_idx.p, h->stage.p, pp.l_idx.size(), hipMemcpyHostToDevice, h->stream));
      off = (pp.l_idx.size() + 63) & ~(size_t)63;
    }
    for (int i = 0; i < NA; ++i) {
      if (!src[i] || src[i]->empty()) continue;
      const size_t bytes = src[i]->size() * sizeof(double);
      HIP_OK(dst[i]->ensure(src[i]->size()));
      memcpy(h->stage.p + off, src[i]->data(), bytes);
      HIP_OK(hipMemcpyAsync(dst[i]->p, h->stage.p + off, bytes, hipMemcpyHostToDevice, h->stream));
      off += bytes;
    }
  }
  HIP_OK(h->d_fns.ensure(pp.fns.size()));
  HIP_OK(hipMemcpyAsync(h->d_fns.p, pp.fns.data(), sizeof(TimeFn) * pp.fns.size(), hipMemcpyHostToDevice, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  if (timing) {
    auto t2 = std::chrono::steady_clock::now();
    fprintf(stderr, "[dfx] set_params: pack %.2f ms, upload %.2f ms\n", std::chrono::duration<double, std::milli>(t1 - t0).count(),
            std::chrono::duration<double, std::milli>(t2 - t1).count());
  }
  h->have_params = true;
  h->have_traj = false;
  h->have_fields = false;
  return 0;
}

int dfx_reserve(dfx_handle* h, int64_t max_steps, int32_t max_timepoints, int32_t keep_trajectory) {
  HIP_OK(hipSetDevice(h->device));
  const Plan& pl = h->pl;
  const size_t B = pl.batch, nb = pl.n_blocks, rec = nb * kStep;
  if (ensure_work_buffers(h)) return 2;
  if (ensure_adjoint_buffers(h)) return 2;
  HIP_OK(h->d_fields.ensure(B * (size_t)max_timepoints * nb * 6));
  HIP_OK(h->d_G.ensure(B * (size_t)max_timepoints * nb * 6));
  HIP_OK(h->d_tmp.ensure(B * (size_t)max_timepoints * nb * 6));
  HIP_OK(h->d_target.ensure(nb));
  HIP_OK(h->d_obj.ensure(B));
  HIP_OK(h->d_segs.ensure((size_t)max_timepoints * (2 + (size_t)(max_steps / std::max(1, max_timepoints - 1)) / kMaxGraphSteps)));
  if (keep_trajectory) {
    hipError_t e = h->d_traj.ensure(B * (size_t)(max_steps + 1) * rec);
    if (e != hipSuccess) { h->err = "reserve: cannot allocate the trajectory checkpoint"; return 2; }
    (void)ensure_stage_checkpoint(h, max_steps);
  }
  return 0;
}

int dfx_forward(dfx_handle* h, const double* state0, const double* timepoints, int32_t n_timepoints,
                int32_t steps_per_interval, int32_t keep_trajectory, double* fields, dfx_stats* stats) {
  if (n_timepoints < 1 || steps_per_interval < 1) { h->err = "forward: need >= 1 timepoint and >= 1 step per interval"; return 1; }
  std::vector<int32_t> spis((size_t)std::max(0, n_timepoints - 1), steps_per_interval);
  return dfx_forward_grid(h, state0, timepoints, n_timepoints, spis.data(), nullptr, keep_trajectory, fields, stats);
}

int dfx_forward_grid(dfx_handle* h, const double* state0, const double* timepoints, int32_t n_timepoints,
                     const int32_t* steps_per_interval, const double* step_times, int32_t keep_trajectory,
                     double* fields, dfx_stats* stats) {
  HIP_OK(hipSetDevice(h->device));
  if (!h->have_params) { h->err = "forward: set_params first"; return 1; }
  h->adaptive = false;
  if (n_timepoints < 1) { h->err = "forward: need >= 1 timepoint and >= 1 step per interval"; return 1; }
  const Plan& pl = h->pl;
  const size_t B = pl.batch, nb = pl.n_blocks, rec = nb * kStep;
  const int Tn = n_timepoints;
  h->ts.assign(timepoints, timepoints + Tn);
  h->spis.assign(steps_per_interval, steps_per_interval + (Tn - 1));
  h->step0.assign(Tn, 0);
  for (int k = 0; k + 1 < Tn; ++k) {
    if (h->spis[k] < 1) { h->err = "forward: need >= 1 timepoint and >= 1 step per interval"; return 1; }
    h->step0[k + 1] = h->step0[k] + h->spis[k];
  }
  h->n_total = h->step0[Tn - 1];
  h->t_steps.clear();
  if (step_times) {
    h->t_steps.assign(step_times, step_times + h->n_total + 1);
    for (long long n = 0; n < h->n_total; ++n)
      if (!(h->t_steps[n + 1] > h->t_steps[n])) { h->err = "forward: step_times must be strictly increasing"; return 1; }
    for (int k = 0; k < Tn; ++k)
      if (h->t_steps[h->step0[k]] != timepoints[k]) { h->err = "forward: step_times must contain every timepoint at the start of its interval"; return 1; }
    HIP_OK(h->d_tsteps.ensure(h->t_steps.size()));
    HIP_OK(hipMemcpyAsync(h->d_tsteps.p, h->t_steps.data(), sizeof(double) * h->t_steps.size(), hipMemcpyHostToDevice, h->stream));
  }
  if (ensure_work_buffers(h)) return 2;
  HIP_OK(h->d_fields.ensure(B * Tn * nb * 6));
  h->have_traj = false;
  if (keep_trajectory) {
    hipError_t e = h->d_traj.ensure(B * (size_t)(h->n_total + 1) * rec);
    if (e != hipSuccess) {
      h->err = "forward: cannot allocate the trajectory checkpoint (" + std::to_string((B * (h->n_total + 1) * rec * 8) >> 20) + " MiB)";
      return 2;
    }
    h->have_traj = true;
    h->dense = ensure_stage_checkpoint(h, h->n_total);
  }
  build_segments(h);
  HIP_OK(h->d_segs.ensure(std::max<size_t>(1, h->segs.size())));
  if (!h->segs.empty())
    HIP_OK(hipMemcpyAsync(h->d_segs.p, h->segs.data(), sizeof(Seg) * h->segs.size(), hipMemcpyHostToDevice, h->stream));
  std::vector<int> cursors(2 + 64, -1);   // [0] unused, [1] non-finite flag, [2+g] segment cursor of group g
  cursors[1] = 0;
  HIP_OK(hipMemcpyAsync(h->d_seg_idx.p, cursors.data(), cursors.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
  HIP_OK(hipMemcpyAsync(h->d_state0.p, state0, sizeof(double) * B * nb * 6, hipMemcpyHostToDevice, h->stream));
  DevCtx c = make_ctx(h);
  h->launches = 0;
  hipLaunchKernelGGL(k_init, slot_grid(h), dim3(kThreads), 0, h->stream, c, (const double*)h->d_state0.p, timepoints[0], 0);
  if (c.traj)
    hipLaunchKernelGGL(k_checkpoint0, dim3((unsigned)((rec + kThreads - 1) / kThreads), (unsigned)B), dim3(kThreads), 0, h->stream, c);
  dim3 g3((unsigned)((nb * 3 + kThreads - 1) / kThreads), (unsigned)B);
  hipLaunchKernelGGL(k_snapshot, g3, dim3(kThreads), 0, h->stream, c, h->d_fields.p, 0, h->d_seg_idx.p + 1);
  HIP_OK(hipEventRecord(h->ev0, h->stream));
  if (fork_groups(h)) return 2;
  for (size_t si = 0; si < h->segs.size(); ++si) {
    const Seg& sg = h->segs[si];
    for (int gi = 0; gi < (int)h->groups.size(); ++gi) {
      if (int rc = run_segment(h, c, gi, sg.n_steps, 0)) return rc;
      if (sg.j0 + sg.n_steps == h->spis[sg.interval]) {   // buffer 0 holds the state at the end of the interval
        const Group& gr = h->groups[gi];
        hipLaunchKernelGGL(k_snapshot, dim3(g3.x, gr.nm), dim3(kThreads), 0, gr.stream, group_ctx(h, c, gi), h->d_fields.p, sg.interval + 1,
                           h->d_seg_idx.p + 1);
      }
    }
  }
  if (join_groups(h)) return 2;
  HIP_OK(hipEventRecord(h->ev1, h->stream));
  if (fields) HIP_OK(hipMemcpyAsync(fields, h->d_fields.p, sizeof(double) * B * Tn * nb * 6, hipMemcpyDeviceToHost, h->stream));
  int bad = 0;
  HIP_OK(hipMemcpyAsync(&bad, h->d_seg_idx.p + 1, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  HIP_OK(hipGetLastError());
  h->have_fields = true;
  if (bad) {
    h->have_traj = false;
    h->err = "forward: non-finite state at output " + std::to_string(bad - 1) + " (unstable step size or contact blow-up)";
    return 3;
  }
  if (stats) {
    memset(stats, 0, sizeof(*stats));
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, h->ev0, h->ev1);
    stats->steps = h->n_total;
    stats->rhs_evals = h->n_total * pl.tab.s;
    stats->launches = h->launches;
    stats->kernel_ms = ms;
    stats->streams = (int64_t)h->groups.size();
    stats->stage_kernel_us = h->n_total ? 1e3 * ms / (double)(h->n_total * pl.tab.s) : 0.0;
    stats->stage_checkpoint = c.AD ? 1 : 0;
  }
  return 0;
}


// ---- adaptive forward (reference odeint semantics) ------------------------------------------------
int dfx_forward_adaptive(dfx_handle* h, const double* state0, const double* timepoints, int32_t n_timepoints,
                         double rtol, double atol, int64_t max_attempts, double* fields, dfx_stats* stats) {
  HIP_OK(hipSetDevice(h->device));
  if (!h->have_params) { h->err = "forward_adaptive: set_params first"; return 1; }
  if (n_timepoints < 1) { h->err = "forward_adaptive: need >= 1 timepoint"; return 1; }
  if (h->pl.tab.s != 6) { h->err = "forward_adaptive: the adaptive controller is defined for the dopri5 tableau"; return 1; }
  const Plan& pl = h->pl;
  const size_t B = pl.batch, nb = pl.n_blocks, nd = nb * 3;
  const int Tn = n_timepoints;
  const Dopri D = make_dopri();
  h->ts.assign(timepoints, timepoints + Tn);
  h->spis.clear(); h->n_total = 0;
  h->have_traj = false; h->have_fields = false;
  h->adaptive = true; h->rtol = rtol; h->atol = atol;
  h->have_adaptive_record = false;
  if (ensure_work_buffers(h)) return 2;
  const int n_wg = (pl.n_slots + kThreads - 1) / kThreads;
  const int n_partials = (pl.n_slots + 63) / 64;
  HIP_OK(h->d_fields.ensure(B * Tn * nb * 6));
  HIP_OK(h->d_clock.ensure(B));
  HIP_OK(h->d_err_partial.ensure(B * n_wg * 4));
  h->n_counts = std::max(0, Tn - 1);
  HIP_OK(h->d_step_counts.ensure(std::max<size_t>(1, B * h->n_counts)));
  HIP_OK(hipMemsetAsync(h->d_step_counts.p, 0, sizeof(int) * std::max<size_t>(1, B * h->n_counts), h->stream));
  HIP_OK(h->d_acc_times.ensure(B * (size_t)kAccCap));
  HIP_OK(h->d_ts.ensure(Tn));
  HIP_OK(h->d_tmp.ensure(std::max<size_t>(B * nb * 6, B)));
  HIP_OK(hipMemcpyAsync(h->d_ts.p, timepoints, sizeof(double) * Tn, hipMemcpyHostToDevice, h->stream));
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
  hipLaunchKernelGGL(k_init, slot_grid(h), dim3(kThreads), 0, h->stream, c, (const double*)h->d_state0.p, timepoints[0], 0);
  hipLaunchKernelGGL(k_snapshot, g3, dim3(kThreads), 0, h->stream, c, h->d_fields.p, 0, h->d_seg_idx.p + 1);
  launch_fwd(h, c, 0, 0, 0, -1, 0, 0);                      // A_0 = f(y0, t0)
  std::vector<double> A((size_t)B * 7 * nd), V0(B * 7 * nd);
  HIP_OK(hipMemcpyAsync(A.data(), h->d_A.p, sizeof(double) * A.size(), hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  // initial step size (Hairer II.4 as restated by jax, order 4), per member
  std::vector<double> y1(B * 2 * nd), tm(B), h0(B), d1v(B);
  for (size_t m = 0; m < B; ++m) {
    const double* q = state0 + m * 2 * nd; const double* v = q + nd; const double* a = A.data() + m * 7 * nd;
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
  HIP_OK(hipMemcpyAsync(A.data(), h->d_A.p, sizeof(double) * A.size(), hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  d_tm.release();
  for (size_t m = 0; m < B; ++m) {
    const double* q = state0 + m * 2 * nd; const double* v = q + nd;
    const double* a0 = A.data() + m * 7 * nd; const double* a1 = a0 + nd; const double* v1 = y1.data() + m * 2 * nd + nd;
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
  auto enqueue_attempt = [&]() {
    // evaluations at S_1..S_5, candidate y1 into buffer 3, then the FSAL evaluation with the error estimate
    static const int inb[6] = {0, 1, 2, 1, 2, 1}, outb[6] = {0, 2, 1, 2, 1, 3};
    for (int i = 1; i <= 5; ++i) launch_fwd(h, c, i, 0, inb[i], outb[i], 0, 0);
    if (pl.model == kNonlinear) {
      if (pl.contact) hipLaunchKernelGGL((k_fwd_stage<kNonlinear, 1>), slot_grid(h), dim3(kThreads), 0, h->stream, c, sc_err, 6, 0, 3, -1, 0, 2);
      else hipLaunchKernelGGL((k_fwd_stage<kNonlinear, 0>), slot_grid(h), dim3(kThreads), 0, h->stream, c, sc_err, 6, 0, 3, -1, 0, 2);
    } else {
      if (pl.contact) hipLaunchKernelGGL((k_fwd_stage<kLinearized, 1>), slot_grid(h), dim3(kThreads), 0, h->stream, c, sc_err, 6, 0, 3, -1, 0, 2);
      else hipLaunchKernelGGL((k_fwd_stage<kLinearized, 0>), slot_grid(h), dim3(kThreads), 0, h->stream, c, sc_err, 6, 0, 3, -1, 0, 2);
    }
    hipLaunchKernelGGL(k_control, dim3((unsigned)B), dim3(kThreads), 0, h->stream, c, n_partials, 2.0 * (double)n_free, Tn);
    hipLaunchKernelGGL(k_prepare, slot_grid(h), dim3(kThreads), 0, h->stream, c, dc, Tn);
    h->launches += 3;
  };
  HIP_OK(hipEventRecord(h->ev0, h->stream));
  // stage-1 record of the first attempt (accept = 0: nothing to commit)
  hipLaunchKernelGGL(k_prepare, slot_grid(h), dim3(kThreads), 0, h->stream, c, dc, Tn);
  const int kAttemptsPerGraph = 32;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  if (h->use_graph) {
    const long long before = h->launches;
    HIP_OK(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
    for (int a = 0; a < kAttemptsPerGraph; ++a) enqueue_attempt();
    HIP_OK(hipStreamEndCapture(h->stream, &graph));
    h->launches = before;
    HIP_OK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    (void)hipGraphDestroy(graph);
  }
  long long attempts_issued = 0;
  int rc = 0;
  while (true) {
    if (exec) { HIP_OK(hipGraphLaunch(exec, h->stream)); h->launches += 8LL * kAttemptsPerGraph; }
    else for (int a = 0; a < kAttemptsPerGraph; ++a) enqueue_attempt();
    attempts_issued += kAttemptsPerGraph;
    HIP_OK(hipMemcpyAsync(clk.data(), h->d_clock.p, sizeof(Clock) * B, hipMemcpyDeviceToHost, h->stream));
    HIP_OK(hipStreamSynchronize(h->stream));
    bool all_done = true;
    for (size_t m = 0; m < B; ++m) {
      if (clk[m].state == 2) { h->err = "forward_adaptive: non-finite error estimate (member " + std::to_string(m) + ")"; rc = 3; }
      if (clk[m].state == 3) { h->err = "forward_adaptive: step size underflow (member " + std::to_string(m) + ")"; rc = 3; }
      if (clk[m].state == 0) all_done = false;
    }
    if (rc || all_done) break;
    if (attempts_issued >= max_attempts) { h->err = "forward_adaptive: step budget exceeded"; rc = 4; break; }
  }
  HIP_OK(hipEventRecord(h->ev1, h->stream));
  if (exec) (void)hipGraphExecDestroy(exec);
  if (rc) { h->adaptive = false; return rc; }
  if (fields) HIP_OK(hipMemcpyAsync(fields, h->d_fields.p, sizeof(double) * B * Tn * nb * 6, hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  HIP_OK(hipGetLastError());
  h->have_fields = true;
  h->adaptive = false;
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
    stats->streams = 1;
    stats->stage_kernel_us = att ? 1e3 * ms / (double)(att * 8) : 0.0;
  }
  return 0;
}

// reverse sweep with the output cotangents already in h->d_G
static int run_adjoint(dfx_handle* h, dfx_grads* grads, dfx_stats* stats, bool kinetic, int n_target) {
  const Plan& pl = h->pl;
  const size_t B = pl.batch;
  const int Tn = (int)h->ts.size();
  set_grad_wishes(h, grads);
  DevCtx c = make_ctx(h);
  h->launches = 0;
  if (zero_grad_accumulators(h)) return 2;
  const int nseg = (int)h->segs.size();
  std::vector<int> cursors(64, nseg);
  HIP_OK(hipMemcpyAsync(h->d_seg_idx.p + 2, cursors.data(), cursors.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
  const double h_last = Tn > 1 ? (h->t_steps.empty() ? (h->ts[Tn - 1] - h->ts[Tn - 2]) / h->spis[Tn - 2]
                                                     : h->t_steps[h->n_total] - h->t_steps[h->n_total - 1]) : 0.0;
  HIP_OK(hipEventRecord(h->ev0, h->stream));
  const int wb = (int)((h->n_total * pl.tab.s - 1) & 1);
  hipLaunchKernelGGL(k_adj_begin, slot_grid(h), dim3(kThreads), 0, h->stream, c, h_last, pl.tab.a[pl.tab.s][pl.tab.s - 1], wb);
  if (c.AD && h->n_total > 0) {     // stage checkpoint: the record the first reverse launch reads
    const long long nr = h->n_total - 1;
    const double t_nr = h->t_steps.empty() ? h->ts[Tn - 1] - h_last : h->t_steps[nr];
    hipLaunchKernelGGL(k_rebuild_first, slot_grid(h), dim3(kThreads), 0, h->stream, c, stage_coef(pl.tab, pl.tab.s - 2), pl.tab.s - 1, nr, h_last, t_nr);
    h->launches++;
  }
  if (fork_groups(h)) return 2;
  for (int si = nseg - 1; si >= 0; --si)
    for (int gi = 0; gi < (int)h->groups.size(); ++gi)
      if (int rc = run_segment(h, c, gi, h->segs[si].n_steps, 1)) return rc;
  if (join_groups(h)) return 2;
  if (kinetic) {
    dim3 g((unsigned)((n_target * 3 + 63) / 64), (unsigned)B);
    hipLaunchKernelGGL(k_kinetic_mass_grad, g, dim3(64), 0, h->stream, c, (const double*)h->d_fields.p, (const int32_t*)h->d_target.p, n_target);
  }
  HIP_OK(hipEventRecord(h->ev1, h->stream));
  if (int rc = collect_grads(h, grads, true)) return rc;
  if (stats) {
    memset(stats, 0, sizeof(*stats));
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, h->ev0, h->ev1);
    stats->steps = h->n_total;
    stats->rhs_evals = h->n_total * pl.tab.s;
    stats->launches = h->launches;
    stats->kernel_ms = ms;
    stats->streams = (int64_t)h->groups.size();
    stats->stage_kernel_us = h->n_total ? 1e3 * ms / (double)(h->n_total * pl.tab.s * 2) : 0.0;
    stats->stage_checkpoint = c.AD ? 1 : 0;
  }
  return 0;
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

int dfx_adjoint(dfx_handle* h, const double* fields_bar, dfx_grads* grads, dfx_stats* stats) {
  HIP_OK(hipSetDevice(h->device));
  if (!h->have_traj) { h->err = "adjoint: run forward with keep_trajectory=1 first"; return 1; }
  const Plan& pl = h->pl;
  const size_t B = pl.batch, nb = pl.n_blocks;
  const int Tn = (int)h->ts.size();
  if (ensure_adjoint_buffers(h)) return 2;
  HIP_OK(h->d_G.ensure(B * Tn * nb * 6));
  HIP_OK(h->d_tmp.ensure(B * Tn * nb * 6));
  HIP_OK(hipMemcpyAsync(h->d_tmp.p, fields_bar, sizeof(double) * B * Tn * nb * 6, hipMemcpyHostToDevice, h->stream));
  DevCtx c = make_ctx(h);
  const size_t total = B * Tn * nb * 3;
  hipLaunchKernelGGL(k_pack_G, dim3((unsigned)((total + kThreads - 1) / kThreads)), dim3(kThreads), 0, h->stream, c,
                     (const double*)h->d_tmp.p, h->d_G.p);
  return run_adjoint(h, grads, stats, false, 0);
}

static int upload_targets(dfx_handle* h, const int32_t* target_blocks, int32_t n_target) {
  for (int i = 0; i < n_target; ++i)
    if (target_blocks[i] < 0 || target_blocks[i] >= h->pl.n_blocks) { h->err = "target block out of range"; return 1; }
  HIP_OK(h->d_target.ensure(std::max(1, n_target)));
  HIP_OK(hipMemcpyAsync(h->d_target.p, target_blocks, sizeof(int32_t) * n_target, hipMemcpyHostToDevice, h->stream));
  HIP_OK(h->d_obj.ensure(h->pl.batch));
  return 0;
}

int dfx_objective_kinetic(dfx_handle* h, const int32_t* target_blocks, int32_t n_target, double* objective) {
  HIP_OK(hipSetDevice(h->device));
  if (!h->have_fields) { h->err = "objective: run forward first"; return 1; }
  if (int rc = upload_targets(h, target_blocks, n_target)) return rc;
  DevCtx c = make_ctx(h);
  hipLaunchKernelGGL(k_kinetic, dim3(h->pl.batch), dim3(kThreads), 0, h->stream, c, (const double*)h->d_fields.p,
                     (const int32_t*)h->d_target.p, n_target, (double*)nullptr, h->d_obj.p);
  HIP_OK(hipMemcpyAsync(objective, h->d_obj.p, sizeof(double) * h->pl.batch, hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  return 0;
}

int dfx_adjoint_kinetic(dfx_handle* h, const int32_t* target_blocks, int32_t n_target, dfx_grads* grads, dfx_stats* stats) {
  HIP_OK(hipSetDevice(h->device));
  if (!h->have_traj || !h->have_fields) { h->err = "adjoint_kinetic: run forward with keep_trajectory=1 first"; return 1; }
  const Plan& pl = h->pl;
  const size_t B = pl.batch, nb = pl.n_blocks;
  const int Tn = (int)h->ts.size();
  if (ensure_adjoint_buffers(h)) return 2;
  if (int rc = upload_targets(h, target_blocks, n_target)) return rc;
  HIP_OK(h->d_G.ensure(B * Tn * nb * 6));
  HIP_OK(hipMemsetAsync(h->d_G.p, 0, sizeof(double) * B * Tn * nb * 6, h->stream));
  DevCtx c = make_ctx(h);
  hipLaunchKernelGGL(k_kinetic, dim3(h->pl.batch), dim3(kThreads), 0, h->stream, c, (const double*)h->d_fields.p,
                     (const int32_t*)h->d_target.p, n_target, h->d_G.p, h->d_obj.p);
  return run_adjoint(h, grads, stats, true, n_target);
}

// ---- test hooks ------------------------------------------------------------------------------
static int hook_prepare(dfx_handle* h, const double* y, double t) {
  const Plan& pl = h->pl;
  const size_t B = pl.batch, nb = pl.n_blocks;
  if (!h->have_params) { h->err = "set_params first"; return 1; }
  h->adaptive = false;
  if (ensure_work_buffers(h)) return 2;
  if (ensure_adjoint_buffers(h)) return 2;
  h->have_traj = false;
  h->have_fields = false;
  h->ts.assign(1, t);
  h->n_total = 1;
  h->t_steps.clear();
  Seg sg;
  sg.t_interval = t; sg.h = 0.0; sg.h_prev = 0.0; sg.base_step = 0; sg.j0 = 0; sg.interval = 0; sg.n_steps = 1; sg.pad = 0;
  HIP_OK(hipMemcpyAsync(h->d_cur.p, &sg, sizeof(Seg), hipMemcpyHostToDevice, h->stream));
  HIP_OK(hipMemcpyAsync(h->d_state0.p, y, sizeof(double) * B * nb * 6, hipMemcpyHostToDevice, h->stream));
  DevCtx c = make_ctx(h);
  hipLaunchKernelGGL(k_init, slot_grid(h), dim3(kThreads), 0, h->stream, c, (const double*)h->d_state0.p, t, 0);
  return 0;
}

int dfx_rhs(dfx_handle* h, const double* y, double t, double* dy) {
  HIP_OK(hipSetDevice(h->device));
  if (int rc = hook_prepare(h, y, t)) return rc;
  const Plan& pl = h->pl;
  const size_t B = pl.batch, nb = pl.n_blocks;
  DevCtx c = make_ctx(h);
  launch_fwd(h, c, 0, 0, 0, -1, 0, 0);
  std::vector<double> A(B * (pl.tab.s + 1) * nb * 3), S(B * (2 * pl.tab.s) * nb * 3);
  HIP_OK(hipMemcpyAsync(A.data(), h->d_A.p, sizeof(double) * A.size(), hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipMemcpyAsync(S.data(), h->d_VEL.p, sizeof(double) * S.size(), hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  HIP_OK(hipGetLastError());
  for (size_t m = 0; m < B; ++m)
    for (size_t b = 0; b < nb; ++b)
      for (int d = 0; d < 3; ++d) {
        const int sidx = pl.block_special[b];
        const bool con = sidx >= 0 && ((pl.special[sidx].con_mask >> d) & 1);
        dy[m * nb * 6 + b * 3 + d] = con ? 0.0 : S[m * (2 * pl.tab.s) * nb * 3 + b * 3 + d];
        dy[m * nb * 6 + nb * 3 + b * 3 + d] = A[m * (pl.tab.s + 1) * nb * 3 + b * 3 + d];
      }
  return 0;
}

int dfx_rhs_vjp(dfx_handle* h, const double* y, double t, const double* lam, double* y_bar, dfx_grads* grads) {
  HIP_OK(hipSetDevice(h->device));
  if (int rc = hook_prepare(h, y, t)) return rc;
  const Plan& pl = h->pl;
  const size_t B = pl.batch, nb = pl.n_blocks;
  set_grad_wishes(h, grads);
  DevCtx c = make_ctx(h);
  c.G = nullptr;
  launch_fwd(h, c, 0, 0, 0, -1, 0, 0);
  HIP_OK(h->d_tmp.ensure(B * nb * 6));
  HIP_OK(hipMemcpyAsync(h->d_tmp.p, lam, sizeof(double) * B * nb * 6, hipMemcpyHostToDevice, h->stream));
  if (zero_grad_accumulators(h)) return 2;
  dim3 g3((unsigned)((nb * 3 + kThreads - 1) / kThreads), (unsigned)B);
  hipLaunchKernelGGL(k_seed_vjp, g3, dim3(kThreads), 0, h->stream, c, (const double*)h->d_tmp.p);
  launch_adj(h, c, 0, 0, 0, 0, 1);
  std::vector<double> YB(B * pl.tab.s * nb * 6);
  HIP_OK(hipMemcpyAsync(YB.data(), h->d_YB.p, sizeof(double) * YB.size(), hipMemcpyDeviceToHost, h->stream));
  if (int rc = collect_grads(h, grads, false)) return rc;
  for (size_t m = 0; m < B; ++m)
    for (size_t b = 0; b < nb; ++b)
      for (int d = 0; d < 3; ++d) {
        y_bar[m * nb * 6 + b * 3 + d] = YB[m * pl.tab.s * nb * 6 + b * 6 + d];
        y_bar[m * nb * 6 + nb * 3 + b * 3 + d] = YB[m * pl.tab.s * nb * 6 + b * 6 + 3 + d];
      }
  return 0;
}

int dfx_energy(dfx_handle* h, const double* u, double* energy) {
  HIP_OK(hipSetDevice(h->device));
  const Plan& pl = h->pl;
  const size_t B = pl.batch, nb = pl.n_blocks;
  if (!h->have_params) { h->err = "energy: set_params first"; return 1; }
  // records straight from u (no constraint override: the energy of the configuration as given)
  if (ensure_work_buffers(h)) return 2;
  std::vector<double> S(B * (2 * pl.tab.s) * nb * kPos, 0.0);
  for (size_t m = 0; m < B; ++m)
    for (size_t b = 0; b < nb; ++b) {
      double* r = S.data() + m * (2 * pl.tab.s) * nb * kPos + b * kPos;
      for (int d = 0; d < 3; ++d) r[d] = u[m * nb * 3 + b * 3 + d];
      r[3] = cos(0.5 * r[2]); r[4] = sin(0.5 * r[2]);
    }
  HIP_OK(hipMemcpyAsync(h->d_POS.p, S.data(), sizeof(double) * S.size(), hipMemcpyHostToDevice, h->stream));
  HIP_OK(h->d_tmp.ensure(B * pl.n_slots));
  DevCtx c = make_ctx(h);
  if (pl.model == kNonlinear) {
    if (pl.contact) hipLaunchKernelGGL((k_energy<kNonlinear, 1>), slot_grid(h), dim3(kThreads), 0, h->stream, c, h->d_tmp.p);
    else hipLaunchKernelGGL((k_energy<kNonlinear, 0>), slot_grid(h), dim3(kThreads), 0, h->stream, c, h->d_tmp.p);
  } else {
    if (pl.contact) hipLaunchKernelGGL((k_energy<kLinearized, 1>), slot_grid(h), dim3(kThreads), 0, h->stream, c, h->d_tmp.p);
    else hipLaunchKernelGGL((k_energy<kLinearized, 0>), slot_grid(h), dim3(kThreads), 0, h->stream, c, h->d_tmp.p);
  }
  std::vector<double> e(B * pl.n_slots);
  HIP_OK(hipMemcpyAsync(e.data(), h->d_tmp.p, sizeof(double) * e.size(), hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  HIP_OK(hipGetLastError());
  for (size_t m = 0; m < B; ++m) {
    double acc = 0.0;
    for (int s = 0; s < pl.n_slots; ++s) acc += e[m * pl.n_slots + s];
    energy[m] = acc;
  }
  return 0;
}

}  // extern "C"
