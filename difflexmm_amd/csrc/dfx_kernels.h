// dfx_kernels.h -- device side of libdfx (gfx950): the structures passed to the kernels and the kernels themselves.
// Included by dfx_engine.hip only (everything lives in an anonymous namespace).
//
//   k_fwd_stage     one Runge-Kutta stage: ligament + contact forces of every unit, stage acceleration, next stage record
//   k_adj_stage     one reverse stage: Hessian-vector product and parameter-gradient contributions from a dual-number
//                   evaluation of the same forces, reverse stage combination; optionally rebuilds a stage record
//   k_control / k_prepare            adaptive step controller and dense output (jax.experimental.ode semantics)
//   k_init, k_snapshot, k_checkpoint0, k_tick, k_adj_begin, k_rebuild_first, k_pack_G, k_kinetic*, k_energy
//                                    small elementwise helpers around the two stage kernels
//
// Lane mapping, data layout, addressing and the load-batch discipline are described in DESIGN.md section 3.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "dfx_stage.h"

using namespace dfx;

// occupancy hints for the two stage kernels (waves per SIMD the register allocator must leave room for).
// Forward: 5 waves (96 VGPRs + 100 B/lane of scratch) measured +7 % forward-only at 16 members per GPU over the
// allocator's own choice (120 VGPRs, 4 waves), neutral at 1..8 members; 6 waves lose it again to spills.  Reverse: any
// forced occupancy spills into the load batch (-20..25 %), left to the allocator (126 VGPRs, 4 waves).
#ifndef DFX_FWD_OCC
#define DFX_FWD_OCC __attribute__((amdgpu_waves_per_eu(5)))
#endif
#ifndef DFX_ADJ_OCC
#define DFX_ADJ_OCC
#endif
#ifndef DFX_ADJ_RB_OCC
#define DFX_ADJ_RB_OCC __attribute__((amdgpu_waves_per_eu(4)))
#endif

namespace dfx {    // types and constants the host side's translation units share (everything device-side stays in the anonymous namespace below)

// Index arithmetic inside the stage kernels is 32-bit (one s_mul / v_mad instead of a 64-bit multiply chain per array);
// dfx_create refuses ensembles whose largest per-handle array would not fit (check_index_range).  The trajectory
// checkpoint and the cotangent table keep 64-bit offsets.
typedef unsigned u32;

#ifndef DFX_THREADS
#define DFX_THREADS 128
#endif
constexpr int kThreads = DFX_THREADS;          // workgroup size of every kernel (128 / 512 measured: profiles/r02_workgroup_size.txt)
constexpr int kWavesPerWg = kThreads / 64;
constexpr int kAccCap = 1 << 20;   // accepted step times recorded per member (adaptive)
constexpr int kMaxGraphSteps = 256;
constexpr int kMaxGroups = 64;   // member groups (one stream each): size of the cursor tables and stride of the graph-cache key
constexpr int kPos = 4;   // doubles per unit position record: x y th sin(th/2)   (two 16-byte chunks; cos(th/2) is derived: half_cos)
constexpr int kStep = kPos + 3;  // doubles per unit in a trajectory checkpoint: position record + velocity (3)

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
  double cur[kMaxStages + 1];  // cur[j] = a[j][i] for j in i+1..s-1, cur[s] = b_i: Kbar_q of THIS stage is recomputed from lambda and the
                               // later stages' Ybar (already loaded for the next stage's Kbar) instead of being stored and re-read
  double c_i;
};

// The adaptive solve that keeps its accepted steps for the reverse sweep (dfx_forward_adaptive_keep): what the controller records per
// member beside the stage records it leaves in the trajectory checkpoint (record r of step n of member m, n = the member's own count of
// accepted steps), and what the reverse stage reads to send the outputs' cotangents through the dense output:
//   t_steps[m][n]   start of step n (n = 0 .. N_m), and once more at N_m + 1: the "step of size zero" whose only launch is the extra
//                   evaluation at the final state (the FSAL slope of the last step)
//   out_ptr[m][n]   first output that lies inside step n or later: the outputs inside step n are [out_ptr[n], out_ptr[n+1])
//   theta[m][k]     relative position of output k inside its step;  dw[m][k][0..6] = the weights B_j(theta) of the stage slopes
//                   (dopri_dense_weights; k_dense_weights fills them before a sweep)
struct AdaptRec {
  double* t_steps;       // batch * stride
  int* out_ptr;          // batch * stride
  double* theta;         // batch * n_out
  long long stride;
};
struct DenseCtx {
  const int* out_ptr;    // batch * stride
  const double* dw;      // batch * n_out * 8
  const int* n_acc;      // batch: accepted steps N_m
  long long stride;
  int n_out, pad;
};

// arguments of the adaptive controller inside the persistent stage loop (dfx_persist_dense.h)
struct AdaptLoopCoef {          // Dormand-Prince with embedded error and dense output (dfx_physics.h: Dopri), acceleration form
  double a[7][7], aa[7][7], e[7], ee[7], cm[7], cma[7], c[7];
};
struct AdaptLoopArgs {
  double* err;                  // 3 * batch * waves_per_member per-wave partials of the squared error ratio (poison = not yet written)
  AdaptRec ar;                  // t_steps == nullptr: the steps are not kept
  double two_n_free;
  long long cap;                // steps the kept-step buffers hold per member
  int n_timepoints, keep;
};


struct DevCtx {
  int n_blocks, n_slots, n_fns, batch, s, n_special, k_uniform, n_timepoints;
  int m0, nbuf;           // first member of the group this launch integrates (one stream per group); stage buffers per member
  int pred[4];            // guessed partner slot = own slot + pred[node slot]
  int ablate, n_wg;       // ablate: profiling experiments on k_fwd_stage (results are wrong when non-zero).  Always 0 in the production library:
                          // only a build with -DDFX_ABLATE reads it from the environment (make_ctx).  The three tests of this always-zero field
                          // stay in the kernel on purpose: compiled out, the register allocator spills 12 scalar registers into the hot path
                          // (forward launch 18.4 -> 19.1 us, profiles/r04_write_through_stores.txt);  n_wg: workgroups per member
  int n_wg3;              // ... of the launches that pack 3-node blocks densely (lane_pos<3>: 40 blocks per 128-thread workgroup)
  int rps;                // records per step in traj: 1 = the step states, s = every stage record (records checkpoint)
  int lam_pairs;          // layout of LAM / YB: 1 = (q, v) of one DOF side by side (b*6 + 2d, + 1), one 16-B access per lane; 0 = (q0 q1 q2 v0 v1 v2)
                          // -- the REBUILD builds of the reverse stage (stage checkpoint, per-ligament gradients) sit at their register limit
                          // and lose the fourth wave per SIMD with 16-B accesses (profiles/r02_wide_per_dof_accesses.txt)
  const int32_t* slot_info;
  const int32_t* block_special;
  const dfx_special* special;
  // extra ligaments of nodes that carry more than one (Plan::ovf_*; null / 0 for every lattice the reference generates)
  const int32_t* ovf_ptr;   // n_slots + 1
  const int32_t* ovf_info;  // n_ovf: 2 * partner slot + end bit
  const double* ovf_p;      // batch * n_ovf * kOvfParams
  double* ovf_g;            // batch * n_ovf * kOvfG: d/d(own void angle | lx ly ks ksh kr am ac kc on end-0 entries)
  int n_ovf, pad_ovf;
  // per-member parameter images, rows indexed by the lane's own slot / DOF (coalesced)
  const double* p_r;      // n_slots*2   own centroid->node vector
  const double* p_l;      // n_slots*2   reference vector of the slot's ligament
  const double* p_k;      // n_slots*4   stiffnesses (only read when they differ between ligaments)
  const double* p_c;      // n_blocks*2   block centroids (distance-based contact only)
  const double* p_phi;    // n_slots*2   the two undeformed void angles (phi1, phi2) of the slot's ligament
  const uint8_t* p_lidx;  // n_slots     index of the slot's reference vector in l_dict (when l_dict_on)
  const double* l_dict;   // 256*4   lx ly |l0| 1/|l0|
  int l_dict_on, damping_uniform;
  int l_dict_lds;         // the dictionary has at most kDictLds entries in every member: every wave copies it into LDS with its first batch of
                          // loads and looks its entries up there (the global lookup was a second, dependent memory round trip)
  const double* cst;      // 16           min_angle cutoff_angle k_contact | uniform k_stretch k_shear k_rot
  const double* inv_m;    // n_blocks*3
  const double* damping;  // n_blocks*3
  const TimeFn* fns;
  const double* fn_tab;   // values of the time functions at every stage time of the segment being replayed (k_fn_table), or null: the
                          // lanes of driven / loaded blocks evaluate them themselves (adaptive steps, test hooks)
  const Seg* cur;         // the segment being replayed
  Clock* clock;           // per-member clocks (adaptive mode) or null
  double* err_partial;    // batch * n_wg * kWavesPerWg per-wave partial sums of the squared error ratio
  int* step_counts;       // batch * (n_timepoints-1) accepted steps per output interval (adaptive)
  double* acc_times;      // batch * acc_cap end times of the accepted steps (adaptive)
  int acc_cap;
  const double* t_steps;  // n_total+1 step boundaries of a caller-chosen grid, or null: equal steps (Seg.h)
  long long ts_stride;    // elements between members in t_steps: 0 = one grid for all, n_total+1 = every member its own (dfx_forward_grid_members)
  double* AD;             // stage checkpoint: batch * (N * s * n_dof) stage accelerations of EVERY step, or null
  long long ad_stride;    // elements between members in AD
  const double* ts_dev;   // output times (adaptive mode)
  double* fields_dev;     // batch * T * n_blocks*6 (adaptive mode writes its dense output here)
  double rtol, atol;
  // state: nbuf = 2s stage buffers per member (two sets for the two-chain reverse sweep); buffer 0 = current step state
  double* traj;           // checkpoints, record-major: [record ordinal][member][POS n_blocks*kPos | VEL n_blocks*3] -- all members' records of
                          // one (step, stage) are contiguous: a launch streams ONE region however many small members it integrates
  double* POS;            // batch * nbuf * n_blocks*kPos
  double* VEL;            // batch * nbuf * n_blocks*3
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
  int* touch;             // touch[0] = 1 once any lane of the sweep has added to the void-angle accumulator (else its download is skipped)
  double* blk_m;          // batch * n_blocks*3    d/d(inertia)
  double* g_c;            // batch * n_blocks*2    d/d(block_centroids) (distance-based contact only)
  int n_npb, pad_npb;     // nodes per block (3 or 4)
  double* blk_c;          // batch * n_blocks*3    d/d(damping) or null
  double* fn_g;           // batch * n_special*MAX_FNS*FN_PARAMS or null
};

}  // namespace dfx

namespace {

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

// ---- lanes of one block.  NPB = 4: a quad (quads; 3-node blocks with their fourth lane idle).  NPB = 3: 3-node blocks packed
// densely -- five triangles in the 15 low lanes of every 16-lane DPP row, the sixteenth lane plays the padding slot of the row's last
// triangle (no ligament, no DOF: it never stores) -- 20 blocks per wave instead of 16.  Memory keeps 4 slots per block either way, so
// kernels of both mappings read and write the same arrays.  Lane moves inside a row: row_shl:n (lane i reads lane i + n), row_shr:n
// (lane i reads lane i - n); what a lane reads from outside its triangle is never selected.
struct LanePos { int slot, b, k; bool valid; };
template <int NPB>
__device__ __forceinline__ LanePos lane_pos(int lwg, int n_blocks) {
  LanePos p;
  const int gtid = lwg * kThreads + (int)threadIdx.x;
  if (NPB == 4) { p.slot = gtid; p.b = gtid >> 2; p.k = gtid & 3; p.valid = p.b < n_blocks; return p; }
  const int j = gtid & 15;
  const int tri = min((j * 11) >> 5, 4);            // j / 3 for j < 15; lane 15 joins triangle 4 as its padding lane (k = 3)
  p.k = j - 3 * tri;
  p.b = (gtid >> 4) * 5 + tri;
  p.valid = p.b < n_blocks;
  p.slot = p.b * 4 + p.k;
  return p;
}
// value held by lane J of the own block
template <int NPB, int J>
__device__ __forceinline__ double blk_bcast(double v, int k) {
  if (NPB == 4) return quad_bcast<J>(v);
  if (J == 0) { const double m1 = dpp_mov<0x111>(v), m2 = dpp_mov<0x112>(v); return k == 0 ? v : (k == 1 ? m1 : m2); }
  if (J == 1) { const double p1 = dpp_mov<0x101>(v), m1 = dpp_mov<0x111>(v); return k == 0 ? p1 : (k == 1 ? v : m1); }
  const double p2 = dpp_mov<0x102>(v), p1 = dpp_mov<0x101>(v);
  return k == 0 ? p2 : (k == 1 ? p1 : v);
}
// lane k of a block receives the block's sum of component k: (a0, a1, a2) = this lane's contributions to (x, y, theta)
template <int NPB>
__device__ __forceinline__ double blk_reduce3(double a0, double a1, double a2, int k) {
  if (NPB == 4) {
    a0 = quad_sum(a0); a1 = quad_sum(a1); a2 = quad_sum(a2);
    return k == 0 ? a0 : (k == 1 ? a1 : a2);
  }
  // transposed: every lane hands component (k+1)%3 to the next lane of its triangle and (k+2)%3 to the one after (cyclic)
  const double own = k == 0 ? a0 : (k == 1 ? a1 : a2);
  const double u = k == 0 ? a1 : (k == 1 ? a2 : a0);     // for lane k+1
  const double w = k == 0 ? a2 : (k == 1 ? a0 : a1);     // for lane k+2
  const double u_m1 = dpp_mov<0x111>(u), u_p2 = dpp_mov<0x102>(u);   // from lane k-1 = i - 1 (k >= 1) / i + 2 (k == 0)
  const double w_p1 = dpp_mov<0x101>(w), w_m2 = dpp_mov<0x112>(w);   // from lane k+1 = i + 1 (k <= 1) / i - 2 (k == 2)
  return own + (k == 0 ? u_p2 : u_m1) + (k == 2 ? w_m2 : w_p1);
}

// XCD-aware workgroup order: hardware deals workgroups round-robin over the 8 XCDs (id % 8 share an L2);
// give every XCD one contiguous band of the lattice so neighbour gathers mostly hit that XCD's own L2.
__device__ __forceinline__ int logical_wg(int bid, int n_wg) {
  const int x = bid & 7, q = bid >> 3, per = n_wg >> 3, rem = n_wg & 7;
  return x * per + (x < rem ? x : rem) + q;
}

// buf >= 0: stage buffer `buf`;  buf < 0: record (-1 - buf) of step n in the trajectory checkpoint (record 0 = the step state;
// records 1 .. s-1 exist in the records checkpoint only; record s of step n IS record 0 of step n + 1).
// Addressing is 32-bit up to the last multiply (round 4): these three functions are inlined five times into a stage kernel, and with
// 64-bit operands throughout each copy was a chain of ~30 scalar instructions (five 64 x 64 multiplies) in every wave's prologue --
// ~150 of the 320 scalar instructions of a forward wave, on the one scalar unit a compute unit has.  The record ordinal fits 32 bits
// (choose_checkpoint refuses a checkpoint of 2^32 records or more), stage-buffer element indices fit 31 bits (check_index_range).
__device__ __forceinline__ double* traj_rec(const DevCtx& c, int m, int buf, long long n) {
  const u32 ord = ((u32)n * (u32)c.rps + (u32)(-1 - buf)) * (u32)c.batch + (u32)m;
  return c.traj + (size_t)ord * (size_t)((u32)c.n_blocks * (u32)kStep);
}
__device__ __forceinline__ const double* pos_in(const DevCtx& c, int m, int buf, long long n) {
  if (buf >= 0) return c.POS + (size_t)(((u32)m * (u32)c.nbuf + (u32)buf) * (u32)c.n_blocks * (u32)kPos);
  return traj_rec(c, m, buf, n);
}
__device__ __forceinline__ const double* vel_in(const DevCtx& c, int m, int buf, long long n) {
  if (buf >= 0) return c.VEL + (size_t)(((u32)m * (u32)c.nbuf + (u32)buf) * (u32)c.n_blocks * 3u);
  return traj_rec(c, m, buf, n) + (size_t)((u32)c.n_blocks * (u32)kPos);
}

__global__ void k_tick(const Seg* segs, int* seg_idx, int delta, Seg* cur) {
  int i = *seg_idx + delta;
  *seg_idx = i;
  *cur = segs[i];
}
// the same without a cursor: segment i, chosen by the host (segments checkpoint: intervals are revisited out of order)
__global__ void k_set_seg(const Seg* segs, int i, Seg* cur) { *cur = segs[i]; }

struct TimeVals { double g, gt; };
// step boundaries of member m (caller-chosen grids; one table for all members or one per member)
__device__ __forceinline__ const double* steps_of(const DevCtx& c, int m) { return c.t_steps + (size_t)m * (size_t)c.ts_stride; }

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

// ---- time functions, tabulated per segment ------------------------------------------------------------------------------------
// Every lane of a driven or loaded block needs g_f(t) at the stage time of its launch (and, constrained DOFs, at the next stage
// time for the record it publishes): a dependent load of the function's parameters, a sin / cos pair in software, a second
// evaluation -- a few hundred instructions and three memory round trips on a handful of lanes.  Invisible when launches fill the
// chip; 2.8 of the 7.1 us of a launch-bound forward stage (one 128x128 system: profiles/LABNOTES.md, "Time functions tabulated per segment"), because a
// launch ends with its slowest workgroup.  All those lanes ask for the same numbers, so one small launch per segment computes them
// for every (step, stage time, function, member) and the stage kernels read them with scalar loads issued at the top of the kernel.
//   entry (m, j, r, f): 8 doubles = g, dg/dt, dg/dp[0..4], pad;  r = 0 .. s: stage times t_n + c_r h with c_s = 1
constexpr int kFnEntry = 8;
constexpr int kFnRows = kMaxStages + 1;
__device__ __forceinline__ const double* fn_tab_row(const DevCtx& c, int m, int j, int r) {
  return c.fn_tab + (((size_t)m * kMaxGraphSteps + j) * kFnRows + r) * (DFX_MAX_FNS * kFnEntry);
}
// A table row has a uniform address, so the compiler reads it with scalar loads -- into scalar registers, of which the stage kernels
// have none to spare (106 in use: the first version of this table cost the main path 60 v_readlane / v_writelane spills per wave, +8 %
// of its instructions, found in the SQ counters).  An offset the compiler cannot prove uniform makes them ordinary vector loads.
__device__ __forceinline__ u32 lane_zero() {
  u32 z;
  asm volatile("v_mov_b32 %0, 0" : "=v"(z));
  return z;
}
__device__ __forceinline__ double fn_tab_get(const double* row, int f, int e, u32 z) {
  return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(row) + ((u32)((f * kFnEntry + e) * 8) + z));
}
struct StageTimes { double c[kFnRows]; };
__global__ __launch_bounds__(64) void k_fn_table(DevCtx c, StageTimes st, int n_steps, double* tab) {
  const int m = blockIdx.y + c.m0;
  const int idx = blockIdx.x * 64 + threadIdx.x;
  if (idx >= n_steps * (c.s + 1) * c.n_fns) return;
  const int f = idx % c.n_fns, r = (idx / c.n_fns) % (c.s + 1), j = idx / (c.n_fns * (c.s + 1));
  const Seg sg = *c.cur;
  const long long n = sg.base_step + j;
  double h = sg.h, t = sg.t_interval + (sg.j0 + j) * sg.h;
  if (c.t_steps) { const double* ts = steps_of(c, m); t = ts[n]; h = ts[n + 1] - t; }
  double g, gt, gp[kMaxFnParams];
  eval_time_fn(c.fns[(u32)m * DFX_MAX_FNS + f], t + st.c[r] * h, g, gt, gp);
  double* e = tab + ((((size_t)m * kMaxGraphSteps + j) * kFnRows + r) * DFX_MAX_FNS + f) * kFnEntry;
  e[0] = g; e[1] = gt;
  for (int i = 0; i < kMaxFnParams; ++i) e[2 + i] = gp[i];
}

// records of a full (2, n_blocks, 3) state at time t0 -> stage buffer `buf`  (constrained DOFs follow c(t0), c'(t0))
//   stride: elements between members in state0 (0: packed (batch, 2, n, 3); a row of the resident (batch, T, 2, n, 3) history otherwise)
//   n0: step ordinal of the state; with a caller-chosen grid its time is read there (every member may have its own)
__global__ __launch_bounds__(kThreads) void k_init(DevCtx c, const double* state0, double t0, int buf, long long stride, long long n0) {
  const int m = blockIdx.y + c.m0;
  if (c.t_steps) t0 = steps_of(c, m)[n0];
  const int tid = blockIdx.x * kThreads + threadIdx.x;
  if (tid >= c.n_slots) return;
  const int b = tid >> 2, d = tid & 3;
  if (d == 3) return;
  const size_t nd = (size_t)c.n_blocks * 3, ms = stride ? (size_t)stride : 2 * nd;
  double q = state0[(size_t)m * ms + b * 3 + d], v = state0[(size_t)m * ms + nd + b * 3 + d];
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
    pr[3] = sn;
  }
}

// fields[m, k] <- (disp, vel) of stage buffer `buf` (or, buf < 0, of the checkpointed state of step n)
//   bad: one word, set when any output row of any member is not finite; bad_m (or null): one word per member (a member that
//   diverges must not take the ensemble with it -- problems/quads_focusing_multi_input.py:66-77 yields NaN for that member only)
__global__ __launch_bounds__(kThreads) void k_snapshot(DevCtx c, double* fields, int k, int* bad, int buf, long long n, int* bad_m) {
  const int m = blockIdx.y + c.m0;
  const int tid = blockIdx.x * kThreads + threadIdx.x;
  if (tid >= c.n_blocks * 3) return;
  const int b = tid / 3, d = tid % 3;
  double* f = fields + ((size_t)m * c.n_timepoints + k) * c.n_blocks * 6;
  const double q = pos_in(c, m, buf, n)[(size_t)b * kPos + d];
  const double v = vel_in(c, m, buf, n)[tid];
  f[tid] = q;
  f[(size_t)c.n_blocks * 3 + tid] = v;
  if (!isfinite(q) || !isfinite(v)) { *bad = k + 1; if (bad_m) bad_m[m] = k + 1; }   // any writer wins: only "some output row is not finite" matters
}

// copy stage buffer 0 into record 0 of step n of the checkpoint (the initial state; the start of an interval in the segments level)
__global__ __launch_bounds__(kThreads) void k_checkpoint0(DevCtx c, long long n) {
  const int m = blockIdx.y + c.m0;
  const int tid = blockIdx.x * kThreads + threadIdx.x;
  if (tid >= c.n_blocks * kStep) return;
  double* t = traj_rec(c, m, -1, n);
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
// Stores.  Plain stores stay dirty in the XCD's L2 until the end-of-kernel release writes them back (nothing survives a kernel
// boundary in the L2s anyway: the next launch's readers sit on other XCDs too); `sc1` stores are written through as they are issued, so
// the launch does not end with a write-back burst (MI355X_MICROARCH.md: "publish-large").  The stage kernels' stores take that form when
// DevCtx::wt is set (launches that fill the chip: 16 x 128x128 forward launch 18.3 -> 17.8 us, reverse 32.9 -> 32.3 us, job +1.5 %;
// a single 128x128 system loses 3 % with it -- profiles/r04_write_through_stores.txt).
typedef unsigned dfx_u4 __attribute__((ext_vector_type(4)));
typedef unsigned dfx_u2 __attribute__((ext_vector_type(2)));
template <class T>
__device__ __forceinline__ void stg_wt(void* base, u32 byte_off, T v) {
  static_assert(sizeof(T) == 16 || sizeof(T) == 8 || sizeof(T) == 4, "write-through store: 4, 8 or 16 bytes");
  if constexpr (sizeof(T) == 16) {
    dfx_u4 x;
    __builtin_memcpy(&x, &v, 16);
    asm volatile("global_store_dwordx4 %0, %1, %2 sc1" ::"v"(byte_off), "v"(x), "s"(base) : "memory");
  } else if constexpr (sizeof(T) == 8) {
    dfx_u2 x;
    __builtin_memcpy(&x, &v, 8);
    asm volatile("global_store_dwordx2 %0, %1, %2 sc1" ::"v"(byte_off), "v"(x), "s"(base) : "memory");
  } else {
    unsigned x;
    __builtin_memcpy(&x, &v, 4);
    asm volatile("global_store_dword %0, %1, %2 sc1" ::"v"(byte_off), "v"(x), "s"(base) : "memory");
  }
}
template <class T>
__device__ __forceinline__ void stg(void* base, u32 byte_off, T v) {
  *reinterpret_cast<T*>(reinterpret_cast<char*>(base) + byte_off) = v;
}
// the stage kernels' own stores (uniform base, uniform choice)
template <class T>
__device__ __forceinline__ void stg_m(int wt /* compile-time constant at every call */, void* base, u32 byte_off, T v) {
  if (wt) stg_wt<T>(base, byte_off, v); else stg<T>(base, byte_off, v);
}
// Streaming variants for data that is touched once per launch and next by a LATER launch (checkpoint records, Ybar, the accumulators'
// read-modify-write): the non-temporal hint keeps them from displacing what the neighbour gathers hit in the L2.  Measured, not
// assumed: DFX_NT is off unless a build defines it (profiles/r03_nontemporal_hint.txt).
typedef double dfx_d2 __attribute__((ext_vector_type(2)));
template <class T> struct NtType { typedef T type; };
template <> struct NtType<double2> { typedef dfx_d2 type; };
template <class T>
__device__ __forceinline__ T ldg_s(const void* base, u32 byte_off) {
#ifdef DFX_NT
  typedef typename NtType<T>::type V;
  const V v = __builtin_nontemporal_load(reinterpret_cast<const V*>(reinterpret_cast<const char*>(base) + byte_off));
  T out;
  __builtin_memcpy(&out, &v, sizeof(T));
  return out;
#else
  return ldg<T>(base, byte_off);
#endif
}
template <class T>
__device__ __forceinline__ void stg_s(void* base, u32 byte_off, T v) {
#if defined(DFX_NT)
  typedef typename NtType<T>::type V;
  V x;
  __builtin_memcpy(&x, &v, sizeof(T));
  __builtin_nontemporal_store(x, reinterpret_cast<V*>(reinterpret_cast<char*>(base) + byte_off));
#else
  stg<T>(base, byte_off, v);
#endif
}

// A pointer-typed field of the DevCtx kernel argument, loaded from the kernel-argument segment AT THE POINT OF USE.  The compiler
// fetches every argument a kernel reads with its first cluster of scalar loads; pointers that only the epilogue needs then occupy
// scalar registers across the whole ligament evaluation -- the reverse stage kernel, at its limit of 102, spilled 17 of them into
// v_writelane / v_readlane pairs in its hot path (34 vector instructions per wave of 630).  DevCtx is the first kernel argument.
template <class T>
__device__ __forceinline__ T* late_arg(u32 byte_offset) {
  T* v;
  asm volatile("s_load_dwordx2 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(__builtin_amdgcn_kernarg_segment_ptr()), "i"(byte_offset) : "memory");
  return v;
}
// (offset inside the kernel-argument segment = offset inside DevCtx: every __global__ function that reaches a DFX_LATE must take DevCtx
// BY VALUE AS ITS FIRST PARAMETER -- k_adj_stage says so with a static_assert on its own signature)
#define DFX_LATE(T, field) late_arg<T>((u32)offsetof(DevCtx, field))
template <class First, class... Rest> struct first_param_is_devctx { static constexpr bool value = std::is_same<First, DevCtx>::value; };
template <class R, class... A> constexpr bool kernel_takes_devctx_first(R (*)(A...)) { return first_param_is_devctx<A...>::value; }

// uniform bases of member m's parameter arrays
struct MemberBases {
  const double *p_r, *p_phi, *p_l, *p_k, *cst, *l_dict, *p_c, *ovf_p;
  const uint8_t* p_lidx;
};
__device__ __forceinline__ MemberBases member_bases(const DevCtx& c, int m) {
  MemberBases B;
  // element offsets in 32 bits (dfx_create refuses handles whose arrays reach 2^31 elements): one s_mul_i32 each instead of a
  // 64-bit product (three multiplies and two adds on the CU's one scalar unit)
  const u32 um = (u32)m, ps = um * (u32)c.n_slots;
  B.p_r = c.p_r + (size_t)(ps * 2); B.p_phi = c.p_phi + (size_t)(ps * 2); B.p_l = c.p_l + (size_t)(ps * 2); B.p_k = c.p_k + (size_t)(ps * 4);
  B.ovf_p = c.ovf_p + (size_t)(um * (u32)c.n_ovf * kOvfParams);
  B.cst = c.cst + (size_t)(um * 16); B.l_dict = c.l_dict + (size_t)(um * 1024); B.p_lidx = c.p_lidx + (size_t)ps;
  B.p_c = c.p_c + (size_t)(um * (u32)c.n_blocks * 2);
  return B;
}

struct LaneIn {
  BlockRec<double> o, p;
  double rox, roy, rpx, rpy, lx, ly, l0, il0, ks, ksh, kr, phi1, phi2, am, ac, kc, sgn;
  int info, pslot, guess;
};

struct Partner {
  double2 b0, b1, rp;      // (x, y), (theta, sin(theta/2)), node vector
  double phi;
};

template <int CONTACT>
__device__ __forceinline__ void load_partner(const MemberBases& B, int pslot, const double* POSin, Partner& P) {
  const u32 rec = (u32)(pslot >> 2) * (kPos * 8);
  P.b0 = ldg<double2>(POSin, rec);
  P.b1 = ldg<double2>(POSin, rec + 16);
  P.rp = ldg<double2>(B.p_r, (u32)pslot * 16);
  P.phi = 0.0;      // angle contact: the void angles are loaded by resolve_lane, and only where the contact can engage
}

// Everything a lane needs for its ligament, in two phases so that every load that does not depend on another
// load is in flight before the first wait:
//   issue_lane   own data (one coalesced 16-byte chunk per lane, spread over the quad by DPP later), slot_info, and
//                the partner's data from a GUESSED slot (own slot + the lattice's usual offset for this node slot);
//                the kernels then issue their own epilogue operands;
//   resolve_lane what depends on loaded values: the dictionary entry of the reference vector, and a second gather
//                only for lanes whose real partner is not the guessed one (irregular connectivity).
// Partner data comes from the same arrays the owners read (lines served by the XCD's L2).
constexpr int kDictLds = 16;
struct LaneRaw {
  Partner P;
  double2 pc, ro, lv, dl0, dl1;     // dl0 / dl1: dictionary entry min(lane, kDictLds - 1), on its way to LDS
  double ks, ksh, kr, phi;
  int info, guess, lidx, slot;
};

template <int CONTACT>
__device__ __forceinline__ void issue_lane(const DevCtx& c, const MemberBases& B, int slot, const double* POSin, LaneRaw& R) {
  const int b = slot >> 2, k = slot & 3;      // memory keeps 4 slots per block whatever the lane mapping is
  R.slot = slot;
  R.info = ldg<int>(c.slot_info, (u32)slot * 4);
  R.pc = k < 2 ? ldg<double2>(POSin, ((u32)b * kPos + 2 * k) * 8) : make_double2(0.0, 0.0);
  R.ro = ldg<double2>(B.p_r, (u32)slot * 16);
  // branch-free (a branch here would end the batch of loads): the unused one of the two reads one shared valid address
  R.lidx = (int)ldg<uint8_t>(c.l_dict_on ? (const void*)B.p_lidx : (const void*)B.cst, c.l_dict_on ? (u32)slot : 0u);
  R.lv = ldg<double2>(c.l_dict_on ? B.cst : B.p_l, c.l_dict_on ? 0u : (u32)slot * 16);
  if (c.l_dict_lds) {       // uniform; every lane loads (the entries share one or two cache lines), lanes 0 .. kDictLds-1 will store
    const u32 e = min((u32)(threadIdx.x & 63), (u32)(kDictLds - 1)) * 32;
    R.dl0 = ldg<double2>(B.l_dict, e);
    R.dl1 = ldg<double2>(B.l_dict, e + 16);
  }
  R.ks = R.ksh = R.kr = 0.0;
  if (!c.k_uniform) { R.ks = ldg<double>(B.p_k, (u32)slot * 32); R.ksh = ldg<double>(B.p_k, (u32)slot * 32 + 8); R.kr = ldg<double>(B.p_k, (u32)slot * 32 + 16); }
  R.phi = 0.0;
  const int delta = k == 0 ? c.pred[0] : (k == 1 ? c.pred[1] : (k == 2 ? c.pred[2] : c.pred[3]));   // selects: a dynamic index would be a memory load
  R.guess = min(max(slot + delta, 0), c.n_slots - 1);
  load_partner<CONTACT>(B, R.guess, POSin, R.P);
}

template <int CONTACT, int NPB = 4, int WAVES = kWavesPerWg>
__device__ __forceinline__ void resolve_lane(const DevCtx& c, const MemberBases& B, const double* POSin, LaneRaw& R, LaneIn& L) {
  const int k_ = R.slot & 3;
  const int info = R.info;
  L.info = info;
  double2 lv = R.lv, ln = make_double2(0.0, 0.0);
  if (c.l_dict_lds) {
    // wave-private copy: a wave's LDS operations complete in order, so its own stores are visible to its loads without a barrier
    __shared__ double2 s_dict[WAVES][kDictLds][2];
    double2 (*sd)[2] = s_dict[threadIdx.x >> 6];
    const u32 ln_ = threadIdx.x & 63;
    if (ln_ < (u32)kDictLds) { sd[ln_][0] = R.dl0; sd[ln_][1] = R.dl1; }
    lv = sd[R.lidx][0];
    ln = sd[R.lidx][1];
  } else if (c.l_dict_on) {
    lv = ldg<double2>(B.l_dict, (u32)R.lidx * 32);
    ln = ldg<double2>(B.l_dict, (u32)R.lidx * 32 + 16);
  }
  const int pslot = info < 0 ? R.guess : (info >> 1);
  L.pslot = pslot; L.guess = R.guess;
  if (pslot != R.guess) load_partner<CONTACT>(B, pslot, POSin, R.P);
  const double* cst = B.cst;
  if (c.k_uniform) { L.ks = cst[3]; L.ksh = cst[4]; L.kr = cst[5]; }
  else { L.ks = R.ks; L.ksh = R.ksh; L.kr = R.kr; }
  if (CONTACT) { L.am = cst[0]; L.ac = cst[1]; L.kc = cst[2]; }
  L.o.x = blk_bcast<NPB, 0>(R.pc.x, k_); L.o.y = blk_bcast<NPB, 0>(R.pc.y, k_);
  L.o.th = blk_bcast<NPB, 1>(R.pc.x, k_); L.o.sh = blk_bcast<NPB, 1>(R.pc.y, k_);
  L.o.ch = half_cos(L.o.th, L.o.sh);
  L.p.x = R.P.b0.x; L.p.y = R.P.b0.y; L.p.th = R.P.b1.x; L.p.sh = R.P.b1.y;
  L.p.ch = half_cos(L.p.th, L.p.sh);
  if (CONTACT == 1) {
    // culling bound of pack_params (cst[9], cst[10]): a ligament whose ends have turned against each other by less than kappa_safe
    // cannot touch whatever its undeformed void angles are -- they are not loaded (32 B/unit and one gather), the member's smallest
    // one stands in and yields exact zeros; the others (rare) fetch theirs now, a second round trip for those lanes only
    double2 ph = make_double2(cst[10], cst[10]);
    if (info >= 0 && !(fabs(L.o.th - L.p.th) <= cst[9])) ph = ldg<double2>(B.p_phi, (u32)R.slot * 16);
    L.phi1 = ph.x;
    L.phi2 = ph.y;
  }
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

// ---- extra ligaments of a node (general bond lists) ----------------------------------------------------------------------------
constexpr int kOvfG = 9;
struct OvfLig {
  BlockRec<double> p;
  double rpx, rpy, lx, ly, l0, il0, ks, ksh, kr, phi1, phi2, sgn;
  int info, pslot;
};
__device__ __forceinline__ void load_ovf(const DevCtx& c, const MemberBases& B, int e, const double* POSin, OvfLig& X) {
  X.info = ldg<int>(c.ovf_info, (u32)e * 4);
  X.pslot = X.info >> 1;
  const u32 rec = (u32)(X.pslot >> 2) * (kPos * 8);
  const double2 b0 = ldg<double2>(POSin, rec), b1 = ldg<double2>(POSin, rec + 16);
  const double2 rp = ldg<double2>(B.p_r, (u32)X.pslot * 16);
  const u32 o = (u32)e * (kOvfParams * 8);
  const double2 lv = ldg<double2>(B.ovf_p, o), k01 = ldg<double2>(B.ovf_p, o + 16), k2p = ldg<double2>(B.ovf_p, o + 32);
  X.phi2 = ldg<double>(B.ovf_p, o + 48);
  X.p.x = b0.x; X.p.y = b0.y; X.p.th = b1.x; X.p.sh = b1.y; X.p.ch = half_cos(b1.x, b1.y);
  X.rpx = rp.x; X.rpy = rp.y; X.lx = lv.x; X.ly = lv.y;
  X.l0 = sqrt(lv.x * lv.x + lv.y * lv.y); X.il0 = 1.0 / X.l0;
  X.ks = k01.x; X.ksh = k01.y; X.kr = k2p.x; X.phi1 = k2p.y;
  X.sgn = (X.info & 1) ? 1.0 : -1.0;
}

// ---- distance-based contact (CONTACT == 2; energy.py:222-330): what a lane needs beyond LaneIn ---------------------------------
// Node vectors of the bonded node, its next and its previous node, for the own block (from the neighbouring lanes of the quad: DPP
// rotations, 4- and 3-node blocks) and for the partner block (two more gathers), and the two block centroids.  Not part of the
// single load batch: this variant has no caller in the reference's problems and is written for correctness first.
// EVERY lane of a quad must call it (lanes without a ligament lend their node vector to their neighbours).
struct DistIn {
  double ro[3][2], rp[3][2], cox, coy, cpx, cpy;
};
template <int CTRL4, int CTRL3>
__device__ __forceinline__ double quad_rot(double v, int npb) { return npb == 4 ? dpp_mov<CTRL4>(v) : dpp_mov<CTRL3>(v); }
// value held by the lane of the NEXT node of this lane's node ([1,2,3,0] / [1,2,0,3]) and of the PREVIOUS one ([3,0,1,2] / [2,0,1,3])
__device__ __forceinline__ double from_next(double v, int npb) { return quad_rot<0x39, 0xC9>(v, npb); }
__device__ __forceinline__ double from_prev(double v, int npb) { return quad_rot<0x93, 0xD2>(v, npb); }

__device__ __forceinline__ void load_dist(const DevCtx& c, const MemberBases& B, int slot, const LaneIn& L, DistIn& D) {
  const int b = slot >> 2, n = c.n_npb;
  const int pb = L.pslot >> 2, kp = L.pslot & 3;
  const int kpn = kp + 1 >= n ? 0 : kp + 1, kpp = kp == 0 ? n - 1 : kp - 1;
  const double2 co = ldg<double2>(B.p_c, (u32)b * 16), cp = ldg<double2>(B.p_c, (u32)pb * 16);
  const double2 rpn = ldg<double2>(B.p_r, (u32)(pb * 4 + kpn) * 16), rpp = ldg<double2>(B.p_r, (u32)(pb * 4 + kpp) * 16);
  D.cox = co.x; D.coy = co.y; D.cpx = cp.x; D.cpy = cp.y;
  D.ro[0][0] = L.rox; D.ro[0][1] = L.roy;
  D.ro[1][0] = from_next(L.rox, n); D.ro[1][1] = from_next(L.roy, n);
  D.ro[2][0] = from_prev(L.rox, n); D.ro[2][1] = from_prev(L.roy, n);
  D.rp[0][0] = L.rpx; D.rp[0][1] = L.rpy;
  D.rp[1][0] = rpn.x; D.rp[1][1] = rpn.y;
  D.rp[2][0] = rpp.x; D.rp[2][1] = rpp.y;
}

// ---- forward stage ---------------------------------------------------------------------------
//   in_buf  : stage buffer holding this stage's records, or -1: the checkpoint of step n (reverse recompute, i == 0)
//   out_buf : buffer for the next stage's records (-1: none)
//   y_buf   : 0: step base state (q_n, v_n) in buffer 0;  -1: in the checkpoint of step n
//   write_traj: also store the new record into the checkpoint of step n+1 (last stage, keep_trajectory)
//   NPB: lanes per block (lane_pos); the packed mapping serves 3-node blocks on the fixed grid (not the adaptive controller's error
//   reduction, not the distance-based contact, whose node rotations are quad moves)
//   TAB: the time functions come from the segment's table (k_fn_table) -- a build of its own, so that neither build carries the
//   other's path: with both in one kernel the main path paid 26 scalar-register spills (v_readlane / v_writelane) per wave
//   OVF: the build that also walks the extra ligaments of nodes with more than one (general bond lists) -- a build of its own: the
//   second inlined copy of the ligament arithmetic costs the main path its registers (forward 96 VGPRs + 240 B scratch, reverse 225
//   VGPRs when it sat in the common build), and no lattice the reference generates needs it
//   WT: the stores are written through (stg_m) -- builds of the table kernels for launches that fill the chip (DevCtx::wt)
//   ISTAGE >= 0: the stage index is a compile-time constant (builds of the write-through table kernels, one per stage: which earlier
//   accelerations to load, which coefficients to use and whether this is the last stage are then decided by the compiler, not by
//   ~12 scalar compares and branches per wave)
//   RECS (per-stage builds): 1 = the records are read from and written to the trajectory checkpoint (records level, and the interval
//   re-runs of the segments level): the buffer arguments are constants too; 0 = they are run-time arguments (stage buffers: the first
//   forward pass of the segments level, the state and stages levels)
template <int MODEL, int CONTACT, int NPB = 4, int TAB = 0, int OVF = 0, int WT = 0, int ISTAGE = -1, int RECS = 1>
__global__ __launch_bounds__(kThreads) DFX_FWD_OCC void k_fwd_stage(DevCtx c_arg, StageCoef sc, int i_arg, int j, int in_buf, int out_buf,
                                                        int y_buf, int mode) {
  static_assert(NPB == 4 || CONTACT != 2, "distance-based contact uses the quad mapping");
  const int i = ISTAGE >= 0 ? ISTAGE : i_arg;
  // The per-stage builds are also the builds of the COMMON parameter shape, which the launch code checks before it takes them
  // (hot_shape in dfx_engine.hip): uniform stiffnesses and damping, the reference-vector dictionary in LDS, equal steps, records read
  // from and written to the trajectory checkpoint.  What is a run-time flag in the generic builds is a constant here.
  DevCtx c = c_arg;
  if (ISTAGE >= 0) {
    c.k_uniform = 1; c.l_dict_on = 1; c.l_dict_lds = 1; c.damping_uniform = 1; c.t_steps = nullptr; c.AD = nullptr; c.clock = nullptr;
    if (RECS) { in_buf = -1 - ISTAGE; out_buf = -2 - ISTAGE; y_buf = -1; mode = 0; }
  }
  const int m = blockIdx.y + c.m0;
  const int lwg = logical_wg(blockIdx.x, NPB == 4 ? c.n_wg : c.n_wg3);
  const LanePos lp = lane_pos<NPB>(lwg, c.n_blocks);
  int slot = lp.slot;
  const int write_traj = mode & 1, err_mode = NPB == 4 ? (mode & 2) : 0;
  const bool valid = lp.valid;
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
    // (records kept for the reverse sweep -- dfx_forward_adaptive_keep: the attempt works in the records of step `accepted` of the
    // trajectory checkpoint; a rejected attempt's records are overwritten by the next one)
    sg.t_interval = ck.t; sg.h = ck.h; sg.j0 = 0; sg.base_step = (c.traj && c.rps > 1) ? ck.accepted : 0; j = 0;
  }
  const long long n = sg.base_step + j;
  const u32 nd = (u32)c.n_blocks * 3;
  // time functions at this stage's time and at the next one: read from the segment's table by the few lanes that need them (the row
  // addresses are formed INSIDE their branch: kept live across the kernel they cost scalar registers the main path spills for), or
  // evaluated by those lanes when there is no table
  constexpr bool use_tab = TAB != 0;
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
  double* Am = keep_stages ? c.AD + (size_t)m * c.ad_stride + (size_t)n * ((u32)(c.s - 1) * nd) : c.A + (size_t)((u32)m * (u32)(c.s + 1) * nd);
  const double damp = c.damping_uniform ? B.cst[6 + kd] : ldg<double>(c.damping + (size_t)((u32)m * nd), o_dof);
  const double invm = ldg<double>(c.inv_m + (size_t)((u32)m * nd), o_dof);
  const int sidx = ldg<int>(c.block_special, (u32)b * 4);
  // earlier stage accelerations: all loads issued together (a rolled loop waits for each one in turn)
  double al[kMaxStages - 1];
#pragma unroll
  for (int l = 0; l < kMaxStages - 1; ++l) al[l] = l < i ? ldg<double>(Am + (size_t)l * nd, o_dof) : 0.0;
  LaneIn L;
  resolve_lane<CONTACT, NPB>(c, B, POSin, R, L);
  DistIn D;
  if (CONTACT == 2) load_dist(c, B, slot, L, D);
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
    fth = L.o.th + L.p.th + L.o.ch + L.p.ch + L.o.sh + L.p.sh + (CONTACT == 1 ? L.phi1 + L.phi2 : 0.0);
  } else if (L.info >= 0) {
    BondGrad<double> g;
    bond_grad<MODEL, double>(L.o, L.p, L.rox, L.roy, L.rpx, L.rpy, L.lx, L.ly, L.l0, L.il0, L.ks, L.ksh, L.kr, L.sgn, g);
    fx = g.fx; fy = g.fy; fth = g.fth;
    if (CONTACT == 1) {
      ContactGrad<double> cg;
      contact_grad<double>(L.sgn * (L.o.th - L.p.th), L.phi1, L.phi2, L.am, L.ac, L.kc, cg);
      fth += L.sgn * cg.dkap;
    }
    if (CONTACT == 2) {
      DistContactGrad<double> dg;
      distance_contact_grad<double, double>(L.o, L.p, D.cox, D.coy, D.cpx, D.cpy, D.ro, D.rp, L.info & 1, L.am, L.ac, L.kc, dg);
      fx += dg.fx; fy += dg.fy; fth += dg.fth;
    }
  }
  if (OVF) {              // nodes with more than one ligament: the others, one after the other
    const int e1 = ldg<int>(c.ovf_ptr, (u32)slot * 4 + 4);
    for (int e = ldg<int>(c.ovf_ptr, (u32)slot * 4); e < e1; ++e) {
      OvfLig X;
      load_ovf(c, B, e, POSin, X);
      BondGrad<double> g;
      bond_grad<MODEL, double>(L.o, X.p, L.rox, L.roy, X.rpx, X.rpy, X.lx, X.ly, X.l0, X.il0, X.ks, X.ksh, X.kr, X.sgn, g);
      fx += g.fx; fy += g.fy; fth += g.fth;
      if (CONTACT == 1) {
        ContactGrad<double> cg;
        contact_grad<double>(X.sgn * (L.o.th - X.p.th), X.phi1, X.phi2, L.am, L.ac, L.kc, cg);
        fth += X.sgn * cg.dkap;
      }
    }
  }
  const double dE = blk_reduce3<NPB>(fx, fy, fth, k);
  // ---- DOF epilogue on lanes 0..2
  double h = sg.h, t = sg.t_interval + (sg.j0 + j) * sg.h;
  if (c.t_steps && !c.clock) { const double* ts = steps_of(c, m); t = ts[n]; h = ts[n + 1] - t; }
  double qnext = 0.0, vnext = 0.0;
  if (k < 3) {
    bool constrained = false;
    double fload = 0.0;
    if (sidx >= 0) {
      const dfx_special& sp = c.special[sidx];
      constrained = (sp.con_mask >> k) & 1;
      if (!constrained) {
        if (use_tab) {
          const double* ft_i = fn_tab_row(c, m, j, i);
          const u32 z = lane_zero();
          for (int f = 0; f < c.n_fns; ++f) fload += sp.load_coef[k][f] * fn_tab_get(ft_i, f, 0, z);
        } else {
          double gp[kMaxFnParams];
          for (int f = 0; f < c.n_fns; ++f)
            if (sp.load_coef[k][f] != 0.0) {
              double g, gt;
              eval_time_fn(c.fns[(u32)m * DFX_MAX_FNS + f], t + sc.c_i * h, g, gt, gp);
              fload += sp.load_coef[k][f] * g;
            }
        }
      }
    }
    const double a = constrained ? 0.0 : (fload - dE - damp * v_i) * invm;
    if (!(keep_stages && i == c.s - 1)) stg_m<double>(WT, Am + (size_t)i * nd, o_dof, a);
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
    if (constrained && out_buf != -1) {
      if (use_tab) {
        const dfx_special& sp = c.special[sidx];
        const double* ft_n = fn_tab_row(c, m, j, i + 1);
        const u32 z = lane_zero();
        qnext = 0.0; vnext = 0.0;
        for (int f = 0; f < c.n_fns; ++f) { qnext += sp.con_coef[k][f] * fn_tab_get(ft_n, f, 0, z); vnext += sp.con_coef[k][f] * fn_tab_get(ft_n, f, 1, z); }
      } else {
        TimeVals tv = constrained_value(c, m, c.special[sidx], k, t + sc.c_next * h);
        qnext = tv.g; vnext = tv.gt;
      }
    }
  }
  if (err_mode) {
    // per-wave sum of the squared error ratios (fixed order -> deterministic); lane 0 of each wave stores it
    double r2 = (k < 3 && valid) ? qnext : 0.0;
    for (int off = 32; off > 0; off >>= 1) r2 += __shfl_down(r2, off, 64);
    if ((threadIdx.x & 63) == 0) c.err_partial[((u32)m * c.n_wg + lwg) * kWavesPerWg + (threadIdx.x >> 6)] = r2;
    return;
  }
  if (out_buf == -1) return;
  // ---- publish the next stage record: lanes 0 and 1 each store one aligned 16-byte chunk (x, y) (th, sin th/2)
  const double y1 = blk_bcast<NPB, 1>(qnext, k), th2 = blk_bcast<NPB, 2>(qnext, k);
  double sn, cs;
  fast_sincos(0.5 * th2, &sn, &cs);
  const double2 chunk = k == 0 ? make_double2(qnext, y1) : make_double2(th2, sn);
  if (k < 3 && !(c.ablate & 2)) {
    const u32 o_chunk = ((u32)b * kPos + 2 * k) * 8;
    if (out_buf >= 0) {
      if (k < 2) stg_m<double2>(WT, c.POS + (size_t)(((u32)m * (u32)c.nbuf + (u32)out_buf) * (u32)c.n_blocks * kPos), o_chunk, chunk);
      stg_m<double>(WT, c.VEL + (size_t)(((u32)m * (u32)c.nbuf + (u32)out_buf) * nd), o_dof, vnext);
    }
    if (write_traj || out_buf < -1) {
      // state checkpoint: the new step state, once more; records checkpoint (out_buf < -1): the record goes ONLY there, the
      // next launch reads it from there and so does the reverse sweep
      double* tr = out_buf < -1 ? traj_rec(c, m, out_buf, n) : traj_rec(c, m, -1, n + 1);
      if (k < 2) stg_m<double2>(WT, tr, o_chunk, chunk);
      stg_m<double>(WT, tr + (size_t)c.n_blocks * kPos, o_dof, vnext);
    }
  }
}



// ---- adaptive step control (jax.experimental.ode semantics) --------------------------------------
// one workgroup per member: reduce the per-wave partials in a fixed order, decide, advance the clock
__global__ __launch_bounds__(kThreads) void k_control(DevCtx c, int n_partials, double two_n_free, int n_timepoints, AdaptRec ar) {
  const int m = blockIdx.x + c.m0;
  __shared__ double red[kThreads];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n_partials; i += kThreads) acc += c.err_partial[(size_t)m * c.n_wg * kWavesPerWg + i];
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
    if (ar.t_steps) {       // step n = accepted - 1 is [t_last, t]; the entries of n + 2 stand for the zero-size step after the last one
      const long long n = ck.accepted - 1;
      double* ts = ar.t_steps + (size_t)m * ar.stride;
      int* op = ar.out_ptr + (size_t)m * ar.stride;
      ts[n + 1] = ck.t; ts[n + 2] = ck.t;
      op[n] = ck.out_lo; op[n + 1] = ck.out_hi; op[n + 2] = ck.out_hi;
      for (int kk = ck.out_lo; kk < ck.out_hi; ++kk) ar.theta[(size_t)m * n_timepoints + kk] = (c.ts_dev[kk] - ck.t_last) / (ck.t - ck.t_last);
    }
  }
  ck.h = h_new;
  if (!(h_new > 0.0)) ck.state = 3;
  c.clock[m] = ck;
}

// elementwise: dense output for the outputs crossed by an accepted step, commit (y_n <- y1, k_1 <- k_7), and the
// stage-1 record of the next attempt with the new step size.  cm / cma: mid-point weights (velocity / position form).
struct DenseCoef { double cm[7], cma[7]; double a10; };

//   recs: the step state, the candidate and the stage records live in the trajectory checkpoint (dfx_forward_adaptive_keep): y_n = record 0
//   of step n, the candidate y1 = record 6 of step n = record 0 of step n + 1 (nothing to copy on accept), stage-1 record of the next
//   attempt = record 1 of the step it attempts
__global__ __launch_bounds__(kThreads) void k_prepare(DevCtx c, DenseCoef dc, int n_timepoints, int recs) {
  const int m = blockIdx.y + c.m0;
  const int slot = logical_wg(blockIdx.x, c.n_wg) * kThreads + threadIdx.x;
  if (slot >= c.n_slots) return;
  const Clock ck = c.clock[m];
  if (ck.state) return;
  const int b = slot >> 2, k = slot & 3, kd = k < 3 ? k : 2;
  const size_t nd = (size_t)c.n_blocks * 3;
  const int dof = b * 3 + kd;
  const long long n_new = ck.accepted, n_old = ck.accepted - (ck.accept ? 1 : 0);
  double* POS0 = recs ? traj_rec(c, m, -1, n_old) + (size_t)b * kPos : c.POS + ((size_t)m * c.nbuf + 0) * c.n_blocks * kPos + (size_t)b * kPos;
  double* VEL0 = recs ? traj_rec(c, m, -1, n_old) + (size_t)c.n_blocks * kPos : c.VEL + ((size_t)m * c.nbuf + 0) * nd;
  const double* POS3 = recs ? traj_rec(c, m, -7, n_old) + (size_t)b * kPos : c.POS + ((size_t)m * c.nbuf + 3) * c.n_blocks * kPos + (size_t)b * kPos;
  const double* VEL3 = recs ? traj_rec(c, m, -7, n_old) + (size_t)c.n_blocks * kPos : c.VEL + ((size_t)m * c.nbuf + 3) * nd;
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
      if (!recs) {
        if (k < 2) *reinterpret_cast<double2*>(POS0 + 2 * k) = *reinterpret_cast<const double2*>(POS3 + 2 * k);
        VEL0[dof] = v1;
      }
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
  const double2 chunk = k == 0 ? make_double2(qnext, y1) : make_double2(th2, sn);
  if (k < 3) {
    double* P1 = recs ? traj_rec(c, m, -2, n_new) : c.POS + ((size_t)m * c.nbuf + 1) * c.n_blocks * kPos;
    double* V1 = recs ? traj_rec(c, m, -2, n_new) + (size_t)c.n_blocks * kPos : c.VEL + ((size_t)m * c.nbuf + 1) * nd;
    if (k < 2) *reinterpret_cast<double2*>(P1 + (size_t)b * kPos + 2 * k) = chunk;
    V1[dof] = vnext;
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
    pr[3] = sn;
  }
}

template <int MODEL, int CONTACT>
__global__ __launch_bounds__(kThreads) void k_energy(DevCtx c, double* e_slot) {
  const int m = blockIdx.y + c.m0;
  const int slot = blockIdx.x * kThreads + threadIdx.x;
  if (slot >= c.n_slots) return;
  LaneIn L;
  load_lane<CONTACT>(c, m, slot, pos_in(c, m, 0, 0), L);
  DistIn D;
  if (CONTACT == 2) load_dist(c, member_bases(c, m), slot, L, D);
  double e = 0.0;
  if (L.info >= 0 && !(L.info & 1)) {
    BondGrad<double> g;
    bond_grad<MODEL, double>(L.o, L.p, L.rox, L.roy, L.rpx, L.rpy, L.lx, L.ly, L.l0, L.il0, L.ks, L.ksh, L.kr, L.sgn, g);
    e = g.e;
    if (CONTACT == 1) {
      ContactGrad<double> cg;
      contact_grad<double>(L.sgn * (L.o.th - L.p.th), L.phi1, L.phi2, L.am, L.ac, L.kc, cg);
      e += cg.e;
    }
    if (CONTACT == 2) {
      DistContactGrad<double> dg;
      distance_contact_grad<double, double>(L.o, L.p, D.cox, D.coy, D.cpx, D.cpy, D.ro, D.rp, L.info & 1, L.am, L.ac, L.kc, dg);
      e += dg.e;
    }
  }
  if (c.ovf_ptr) {
    const MemberBases B = member_bases(c, m);
    const int e1 = c.ovf_ptr[slot + 1];
    for (int x = c.ovf_ptr[slot]; x < e1; ++x) {
      OvfLig X;
      load_ovf(c, B, x, pos_in(c, m, 0, 0), X);
      if (X.info & 1) continue;                       // every ligament once: on its end-0 side
      BondGrad<double> g;
      bond_grad<MODEL, double>(L.o, X.p, L.rox, L.roy, X.rpx, X.rpy, X.lx, X.ly, X.l0, X.il0, X.ks, X.ksh, X.kr, X.sgn, g);
      e += g.e;
      if (CONTACT == 1) {
        ContactGrad<double> cg;
        contact_grad<double>(X.sgn * (L.o.th - X.p.th), X.phi1, X.phi2, L.am, L.ac, L.kc, cg);
        e += cg.e;
      }
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
  const double* tr = c.traj + ((size_t)nr * (u32)c.batch + (u32)m) * ((size_t)c.n_blocks * kStep);      // step states only here (rps == 1)
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
  const double2 chunk = k == 0 ? make_double2(qnext, y1) : make_double2(th2, sn);
  if (k < 3) {
    if (k < 2) stg<double2>(c.POS + ((size_t)m * c.nbuf + r) * (u32)c.n_blocks * kPos, ((u32)b * kPos + 2 * k) * 8, chunk);
    stg<double>(c.VEL + ((size_t)m * c.nbuf + r) * nd, o_dof, vnext);
  }
}

// record s-1 of the LAST step, before the reverse sweep starts (every later record is rebuilt by the reverse launch
// that precedes its reader)
__global__ __launch_bounds__(kThreads) void k_rebuild_first(DevCtx c, StageCoef rc, int r, long long nr, double h, double t) {
  const int m = blockIdx.y + c.m0;
  if (c.t_steps) { t = steps_of(c, m)[nr]; h = steps_of(c, m)[nr + 1] - t; }
  const int slot = logical_wg(blockIdx.x, c.n_wg) * kThreads + threadIdx.x;
  if (slot >= c.n_slots) return;
  rebuild_record(c, m, slot >> 2, slot & 3, rc, r, nr, h, t);
}

// ---- reverse stage ---------------------------------------------------------------------------
//   in_buf: stage buffer with the stage records (recomputed), or -1: the checkpoint of step n (i == 0)
//   wbuf_static: >= 0 selects the (w, kbar_q) input buffer (test hook); -1: parity of the stage ordinal
//   BOND_GRADS: also accumulate d/d(reference vector, stiffnesses, contact constants) (only when the caller asks for them:
//   a compile-time switch, the dual parts of those derivatives are dead code otherwise)
template <int MODEL, int CONTACT, int BOND_GRADS, int REBUILD, int NPB = 4, int TAB = 0, int OVF = 0, int WT = 0, int ISTAGE = -1, int DENSE = 0>
//   REBUILD (compile-time: the rebuild code and its registers exist only in the stage-checkpoint build), rb > 0: after its own work the launch rebuilds stage record rb -- of the same step when i >= 2
//   (rb = i - 1, read by the next reverse launch), of the previous step when i == 0 (rb = s - 1); rc = stage_coef(rb - 1)
//   NPB: lanes per block (lane_pos); the packed mapping exists for the records build only
//   DENSE: the reverse stage of an adaptive solve that kept its accepted steps (dfx_forward_adaptive_keep; builds of their own, the others
//   carry none of it): every member has its own number of steps N_m (a launch beyond a member's last step returns at once; the launch
//   (N_m, 0) is the extra evaluation at the final state: a step of size zero), and the cotangents of the outputs enter through the dense
//   output -- an output inside step n adds g to lambda_n and h_n B_j g to Kbar_j, the FSAL slope's share (j = 6) joins Kbar_0 of step n + 1
__device__ __forceinline__ void adj_stage_body(const DevCtx& c_arg, const AdjCoef& ac, int i_arg, int j, int in_buf, int wbuf_static,
                                               int local_only, const StageCoef& rc, int rb, const DenseCtx& dn = DenseCtx{}) {
  // ISTAGE: see k_fwd_stage.  The per-stage builds take the stage index and the PARAMETER shape (uniform stiffnesses / damping, LDS
  // dictionary, equal steps) as constants -- launch 29.7 -> 29.1 us, 112 VGPRs -- but not the buffer / mode arguments: with those
  // folded as well (in_buf, local_only, wbuf, rb, AD, clock) three variants measured 0.6 - 1.5 us SLOWER although they shed more
  // instructions (profiles/r04_scalar_diet.txt): the reverse kernel is bound by vector issue and those builds schedule worse.
  const int i = ISTAGE >= 0 ? ISTAGE : i_arg;
  DevCtx c = c_arg;
  if (ISTAGE >= 0) { c.damping_uniform = 1; c.k_uniform = 1; c.l_dict_on = 1; c.l_dict_lds = 1; c.t_steps = nullptr; }
  static_assert(NPB == 4 || (CONTACT != 2 && !REBUILD && !BOND_GRADS), "the packed mapping serves the records build without distance contact");
  const int m = blockIdx.y + c.m0;
  const LanePos lp = lane_pos<NPB>(logical_wg(blockIdx.x, NPB == 4 ? c.n_wg : c.n_wg3), c.n_blocks);
  const int slot = lp.slot;
  if (!lp.valid) return;
  const int b = slot >> 2, k = slot & 3, kd = k < 3 ? k : 2;
  const Seg sg = *c.cur;
  const long long n = sg.base_step + j;
  if constexpr (DENSE) {
    const long long n_m = dn.n_acc[m];
    if (n > n_m || (n == n_m && i > 0)) return;
  }
  // the reverse sweep visits forward ordinals n*s+i in decreasing order, so the buffer parity alternates
  const int win = wbuf_static >= 0 ? wbuf_static : (int)((n * c.s + i) & 1);
  const u32 nd = (u32)c.n_blocks * 3, nd6 = (u32)c.n_blocks * 6;
  // ---- load phase
  const double* POSin = pos_in(c, m, in_buf, n);
  const MemberBases B = member_bases(c, m);
  LaneRaw R;
  issue_lane<CONTACT>(c, B, slot, POSin, R);
  const int dof = b * 3 + kd;
  // per-lane byte offsets shared by the per-DOF arrays; lambda and Ybar: see DevCtx::lam_pairs (= !REBUILD)
  const u32 o_dof = (u32)dof * 8, o_b6 = REBUILD ? ((u32)b * 6 + kd) * 8 : ((u32)b * 6 + 2 * kd) * 8;
  const double* Win = c.W + (size_t)(((u32)m * 2 + (u32)win) * nd);
  double w_d = (REBUILD || local_only) ? ldg<double>(Win, o_dof) : 0.0;    // records build: own w recomputed below (like Kbar_q)
  // partner's w from the guessed slot (same batch as everything else)
  double wpx, wpy, wpth;
  { const u32 gb = (u32)(R.guess >> 2) * 24;
    if (!REBUILD) { const double2 wxy = ldg<double2>(Win, gb); wpx = wxy.x; wpy = wxy.y; } else { wpx = ldg<double>(Win, gb); wpy = ldg<double>(Win, gb + 8); }
    wpth = ldg<double>(Win, gb + 16); }
  const double v_i = ldg_s<double>(vel_in(c, m, in_buf, n), o_dof);
  // Kbar_q of this stage: recomputed in the epilogue from lambda and the later stages' Ybar (already loaded for the next stage's
  // Kbar) in the records build -- 48 B/unit less than storing and re-reading it; the REBUILD builds (at their register limit, the
  // second coefficient column costs them scalar-register spills: 42 -> 46.6 us) and the single-RHS VJP hook read the stored / seeded one
  double kq_in = (REBUILD || local_only) ? ldg<double>(c.KQ + (size_t)(((u32)m * 2 + (u32)win) * nd), o_dof) : 0.0;
  const double damp = c.damping_uniform ? B.cst[6 + kd] : ldg<double>(c.damping + (size_t)((u32)m * nd), o_dof);
  const double invm = ldg<double>(c.inv_m + (size_t)((u32)m * nd), o_dof);
  const int sidx = ldg<int>(c.block_special, (u32)b * 4);
  double* YBm = c.YB + (size_t)((u32)m * (u32)c.s * nd6);
  double* LAMm = c.LAM + (size_t)((u32)m * nd6);
  double lq = 0.0, lv = 0.0, sq = 0.0, sv = 0.0, sqc = 0.0, svc = 0.0;
  if (!local_only) {
    if (!REBUILD) {
      if (i == 0 || ac.col[c.s] != 0.0 || ac.cur[c.s] != 0.0) { const double2 l2 = ldg_s<double2>(LAMm, o_b6); lq = l2.x; lv = l2.y; }   // b = 0: lambda not needed
      double2 yb[kMaxStages];
#pragma unroll
      for (int jj = 1; jj < kMaxStages; ++jj) {       // all loads issued together
        const bool on = jj > i && jj < c.s;
        yb[jj] = on ? ldg_s<double2>(YBm + (size_t)((u32)jj * nd6), o_b6) : make_double2(0.0, 0.0);
      }
#pragma unroll
      for (int jj = 1; jj < kMaxStages; ++jj) {
        const double cf = i > 0 ? ac.col[jj] : 1.0;
        sq += cf * yb[jj].x;
        sv += cf * yb[jj].y;
        sqc += ac.cur[jj] * yb[jj].x;
        svc += ac.cur[jj] * yb[jj].y;
      }
      // own w = Kbar_v / m of this stage, from the same values (zero on constrained DOFs by itself: their lambda and Ybar are stored
      // as zeros); the neighbours read the copy the previous launch stored -- equal to rounding
      w_d = ((c.t_steps ? steps_of(c, m)[n + 1] - steps_of(c, m)[n] : sg.h) * (ac.cur[c.s] * lv + svc)) * invm;
    } else {
      if (i == 0 || ac.col[c.s] != 0.0) { lq = ldg<double>(LAMm, o_b6); lv = ldg<double>(LAMm, o_b6 + 24); }
      double yq[kMaxStages], yv[kMaxStages];
#pragma unroll
      for (int jj = 1; jj < kMaxStages; ++jj) {
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
  }
  // ---- dense output (DENSE builds): what the outputs inside this step and inside the previous one add to the Kbar of this stage
  // (`own`: the records build recomputes its own Kbar), to the Kbar this launch hands on (`nxt`), and to lambda_n (`gs`, i == 0) --
  // already scaled by the step sizes; zero on constrained DOFs
  double e_own_q = 0.0, e_own_v = 0.0, e_nxt_q = 0.0, e_nxt_v = 0.0, gs_q = 0.0, gs_v = 0.0;
  if constexpr (DENSE) {
    const bool con_k = sidx >= 0 && ((c.special[sidx >= 0 ? sidx : 0].con_mask >> kd) & 1);
    if (!local_only && !con_k) {
      const double* tsm = steps_of(c, m);
      const double h_n = tsm[n + 1] - tsm[n], h_p = n > 0 ? tsm[n] - tsm[n - 1] : 0.0;
      const int* op = dn.out_ptr + (size_t)m * dn.stride;
      const int lo = op[n], hi = op[n + 1], plo = n > 0 ? op[n - 1] : lo;
      const double* dwm = dn.dw + (size_t)m * dn.n_out * 8;
      const u32 o_g = ((u32)b * 6 + kd) * 8;
      for (int kk = lo; kk < hi; ++kk) {            // outputs inside this step
        const double* Gk = c.G + ((size_t)kk * c.batch + m) * (size_t)nd6;
        const double gq = ldg<double>(Gk, o_g), gv = ldg<double>(Gk, o_g + 24);
        const double* w = dwm + (size_t)kk * 8;
        e_own_q += w[i] * gq; e_own_v += w[i] * gv;
        if (i > 0) { e_nxt_q += w[i - 1] * gq; e_nxt_v += w[i - 1] * gv; }
        else { gs_q += gq; gs_v += gv; }
      }
      e_own_q *= h_n; e_own_v *= h_n; e_nxt_q *= h_n; e_nxt_v *= h_n;
      if (i <= 1) {                                 // outputs inside the previous step: its FSAL slope is this step's first slope
        double e6q = 0.0, e6v = 0.0, e5q = 0.0, e5v = 0.0;
        for (int kk = plo; kk < lo; ++kk) {
          const double* Gk = c.G + ((size_t)kk * c.batch + m) * (size_t)nd6;
          const double gq = ldg<double>(Gk, o_g), gv = ldg<double>(Gk, o_g + 24);
          const double* w = dwm + (size_t)kk * 8;
          e6q += w[6] * gq; e6v += w[6] * gv;
          e5q += w[c.s - 1] * gq; e5v += w[c.s - 1] * gv;
        }
        if (i == 0) { e_own_q += h_p * e6q; e_own_v += h_p * e6v; e_nxt_q = h_p * e5q; e_nxt_v = h_p * e5v; }
        else { e_nxt_q += h_p * e6q; e_nxt_v += h_p * e6v; }
      }
      if (i == 0 && n == 0) {                       // the initial state is output 0 (and any output produced before the first step)
        for (int kk = 0; kk < lo; ++kk) {
          const double* Gk = c.G + ((size_t)kk * c.batch + m) * (size_t)nd6;
          gs_q += ldg<double>(Gk, o_g); gs_v += ldg<double>(Gk, o_g + 24);
        }
      }
      if (!REBUILD) w_d += e_own_v * invm;
    }
  }
  LaneIn L;
  resolve_lane<CONTACT, NPB>(c, B, POSin, R, L);
  DistIn D;
  if (CONTACT == 2) load_dist(c, B, slot, L, D);
  if (L.pslot != L.guess) {
    const u32 pb = (u32)(L.pslot >> 2) * 24;
    if (!REBUILD) { const double2 wxy = ldg<double2>(Win, pb); wpx = wxy.x; wpy = wxy.y; } else { wpx = ldg<double>(Win, pb); wpy = ldg<double>(Win, pb + 8); }
    wpth = ldg<double>(Win, pb + 16);
  }
  const double wox = blk_bcast<NPB, 0>(w_d, k), woy = blk_bcast<NPB, 1>(w_d, k), woth = blk_bcast<NPB, 2>(w_d, k);
  // ---- Hessian-vector product + mixed parameter derivatives of this slot
  double hx = 0.0, hy = 0.0, hth = 0.0;
  double ex = 0.0, ey = 0.0, eth = 0.0;   // dE/du of this slot (value parts): gives the stage acceleration without re-reading it
  double d_rx = 0.0, d_ry = 0.0, d_phi = 0.0;   // this launch's contributions to the node-vector / void-angle gradients
  double dn_x = 0.0, dn_y = 0.0, dp_x = 0.0, dp_y = 0.0, d_cx = 0.0, d_cy = 0.0;   // distance contact: next / previous node, centroid
  // The builds that accumulate only what a design reaches (node vectors, void angles, inertia) take the Hessian-vector product written out
  // by hand (bond_hvp / contact_hvp, dfx_physics.h): a third of the multiply-adds of the dual-number evaluation, on a kernel bound by
  // instruction issue.  Per-ligament gradients, the spring models and the distance-based contact keep the dual numbers.
  constexpr bool kHandHvp = !BOND_GRADS && CONTACT != 2 && (MODEL == kNonlinear || MODEL == kLinearized);
  if constexpr (kHandHvp) {
    if (L.info >= 0) {
      BondHvp hv;
      bond_hvp<MODEL>(L.o, L.p, wox, woy, woth, wpx, wpy, wpth, L.rox, L.roy, L.rpx, L.rpy, L.lx, L.ly, L.l0, L.il0, L.ks, L.ksh, L.kr, L.sgn, hv);
      hx = hv.hx; hy = hv.hy; hth = hv.hth;
      ex = hv.fx; ey = hv.fy; eth = hv.fth;
      d_rx = hv.rx; d_ry = hv.ry;
      if (CONTACT == 1) {
        double dk, dke, p1e, p2e;
        contact_hvp(L.sgn * (L.o.th - L.p.th), L.sgn * (woth - wpth), L.phi1, L.phi2, L.am, L.ac, L.kc, dk, dke, p1e, p2e);
        hth += L.sgn * dke;
        eth += L.sgn * dk;
        d_phi = (L.info & 1) ? p2e : p1e;      // both ends hold the same penalty: each accumulates one of the two void-angle derivatives
      }
    }
  } else
  if (L.info >= 0) {
    BlockRec<Dual> o = seed_rec(L.o, wox, woy, woth);
    BlockRec<Dual> p = seed_rec(L.p, wpx, wpy, wpth);
    BondGrad<Dual> g;
    bond_grad<MODEL, Dual>(o, p, L.rox, L.roy, L.rpx, L.rpy, L.lx, L.ly, L.l0, L.il0, L.ks, L.ksh, L.kr, L.sgn, g);
    hx = g.fx.e; hy = g.fy.e; hth = g.fth.e;
    ex = g.fx.v; ey = g.fy.v; eth = g.fth.v;
    ContactGrad<Dual> cg;
    DistContactGrad<Dual> dg;
    if (CONTACT == 1) {
      contact_grad<Dual>(L.sgn * (o.th - p.th), L.phi1, L.phi2, L.am, L.ac, L.kc, cg);
      hth += L.sgn * cg.dkap.e;
      eth += L.sgn * cg.dkap.v;
    }
    // L += w . F = -w . grad E   =>   dL/dp = -eps(dE/dp)
    d_rx = g.rx.e; d_ry = g.ry.e;
    if (CONTACT == 2) {
      distance_contact_grad<Dual, double>(o, p, D.cox, D.coy, D.cpx, D.cpy, D.ro, D.rp, L.info & 1, L.am, L.ac, L.kc, dg);
      hx += dg.fx.e; hy += dg.fy.e; hth += dg.fth.e;
      ex += dg.fx.v; ey += dg.fy.v; eth += dg.fth.v;
      d_rx += dg.r[0][0].e; d_ry += dg.r[0][1].e;
      dn_x = dg.r[1][0].e; dn_y = dg.r[1][1].e; dp_x = dg.r[2][0].e; dp_y = dg.r[2][1].e;
      d_cx = dg.cx.e; d_cy = dg.cy.e;
    }
    // both ends hold the same contact dual: each accumulates one of the two void-angle derivatives (8 B per lane)
    if (CONTACT == 1) d_phi = (L.info & 1) ? cg.p2.e : cg.p1.e;
    if (!(L.info & 1)) {
      if (BOND_GRADS) {
        double* q = c.g_b + ((size_t)m * (u32)c.n_slots + slot) * 8;
        q[0] -= g.lx.e; q[1] -= g.ly.e; q[2] -= g.ks.e; q[3] -= g.ksh.e; q[4] -= g.kr.e;
        if (CONTACT == 1) { q[5] -= cg.am.e; q[6] -= cg.ac.e; q[7] -= cg.kc.e; }
        if (CONTACT == 2) { q[5] -= dg.am.e; q[6] -= dg.ac.e; q[7] -= dg.kc.e; }
      }
    }
  }
  if (CONTACT == 2) {
    // node-vector gradients of the neighbouring nodes go to the lanes that own them: this lane's node is the NEXT node of its
    // previous lane and the PREVIOUS node of its next lane; centroid gradient: sum over the block, lanes 0 / 1 keep x / y
    d_rx += from_prev(dn_x, c.n_npb) + from_next(dp_x, c.n_npb);
    d_ry += from_prev(dn_y, c.n_npb) + from_next(dp_y, c.n_npb);
    d_cx = quad_sum(d_cx);
    d_cy = quad_sum(d_cy);
    if (k < 2) {
      double* gc = c.g_c + ((size_t)m * (u32)c.n_blocks + b) * 2 + k;
      *gc -= k == 0 ? d_cx : d_cy;
    }
  }
  if (OVF) {              // the node's other ligaments (general bond lists; a build of its own, see k_fwd_stage): same dual evaluation
    const int e1 = ldg<int>(c.ovf_ptr, (u32)slot * 4 + 4);
    for (int e = ldg<int>(c.ovf_ptr, (u32)slot * 4); e < e1; ++e) {
      OvfLig X;
      load_ovf(c, B, e, POSin, X);
      const u32 pb = (u32)(X.pslot >> 2) * 24;
      const double xwx = ldg<double>(Win, pb), xwy = ldg<double>(Win, pb + 8), xwth = ldg<double>(Win, pb + 16);
      BlockRec<Dual> o = seed_rec(L.o, wox, woy, woth);
      BlockRec<Dual> p = seed_rec(X.p, xwx, xwy, xwth);
      BondGrad<Dual> g;
      bond_grad<MODEL, Dual>(o, p, L.rox, L.roy, X.rpx, X.rpy, X.lx, X.ly, X.l0, X.il0, X.ks, X.ksh, X.kr, X.sgn, g);
      hx += g.fx.e; hy += g.fy.e; hth += g.fth.e;
      ex += g.fx.v; ey += g.fy.v; eth += g.fth.v;
      d_rx += g.rx.e; d_ry += g.ry.e;
      double* q = c.ovf_g + ((size_t)m * (u32)c.n_ovf + e) * kOvfG;
      ContactGrad<Dual> cg;
      if (CONTACT == 1) {
        contact_grad<Dual>(X.sgn * (o.th - p.th), X.phi1, X.phi2, L.am, L.ac, L.kc, cg);
        hth += X.sgn * cg.dkap.e;
        eth += X.sgn * cg.dkap.v;
        const double dp = (X.info & 1) ? cg.p2.e : cg.p1.e;
        if (dp != 0.0) { q[0] -= dp; c.touch[0] = 1; }
      }
      if (BOND_GRADS && !(X.info & 1)) {
        q[1] -= g.lx.e; q[2] -= g.ly.e; q[3] -= g.ks.e; q[4] -= g.ksh.e; q[5] -= g.kr.e;
        if (CONTACT == 1) { q[6] -= cg.am.e; q[7] -= cg.ac.e; q[8] -= cg.kc.e; }
      }
    }
  }
  // ---- gradient accumulators.  Every address has exactly one writer per launch: plain load-add-store, with the old values of
  // ALL accumulators requested in one batch (issued here, consumed after the epilogue arithmetic) instead of one memory round
  // trip each at the end of the kernel.  (Fire-and-forget L2 atomics would spare the loads but were measured 10-25 % slower:
  // four fp64 atomics per lane saturate the L2 atomic units.)  Lanes without a ligament / constrained DOFs add zero.
  const u32 ms = (u32)m * (u32)c.n_slots;
  double* const blk_c_ = REBUILD ? c.blk_c : DFX_LATE(double, blk_c);        // (the records build: epilogue pointers fetched late, late_arg)
  double* grm = (REBUILD ? c.g_r : DFX_LATE(double, g_r)) + (size_t)(ms * 2);
  double* gpm = c.g_phi + (size_t)ms;
  double* bmm = (REBUILD ? c.blk_m : DFX_LATE(double, blk_m)) + (size_t)((u32)m * nd);
  double* bcm = blk_c_ + (size_t)((u32)m * nd);
  const double2 r_old = ldg_s<double2>(grm, (u32)slot * 16);
  // the void-angle accumulator moves only where a contact is engaged in this stage (d_phi is an exact zero elsewhere, and contacts
  // are rare: 64 B/unit of the launch's traffic otherwise)
  const bool phi_on = CONTACT == 1 && d_phi != 0.0;
  const double p_old = phi_on ? ldg<double>(gpm, (u32)slot * 8) : 0.0;
  const double bm_old = ldg_s<double>(bmm, o_dof);
  const double bc_old = blk_c_ ? ldg<double>(bcm, o_dof) : 0.0;
  const double hw = blk_reduce3<NPB>(hx, hy, hth, k);
  const double dE = blk_reduce3<NPB>(ex, ey, eth, k);
  if (L.info >= 0 || CONTACT == 2) {     // distance contact: a node without a ligament can still be the neighbour of a bonded node
    stg_m<double2>(WT, grm, (u32)slot * 16, make_double2(r_old.x - d_rx, r_old.y - d_ry));
  }
  if (phi_on) { stg_m<double>(WT, gpm, (u32)slot * 8, p_old - d_phi); c.touch[0] = 1; }
  // ---- DOF epilogue
  double h = sg.h, t_n = sg.t_interval + (sg.j0 + j) * sg.h, h_before = (sg.j0 + j) == 0 ? sg.h_prev : sg.h;
  if (c.t_steps) { const double* ts = steps_of(c, m); t_n = ts[n]; h = ts[n + 1] - t_n; h_before = n > 0 ? t_n - ts[n - 1] : 0.0; }
  if (k < 3) {
    bool constrained = false;
    double fload = 0.0;
    if (sidx >= 0) {
      const dfx_special& sp = c.special[sidx];
      constrained = (sp.con_mask >> k) & 1;
      const double t_i = t_n + ac.c_i * h;
      const double* ft = TAB ? fn_tab_row(c, m, j, i) : nullptr;     // tabulated per segment: k_fn_table (a build of its own, see k_fwd_stage)
      double gp[kMaxFnParams];
      for (int f = 0; f < c.n_fns; ++f) {
        const double coef = constrained ? -hw * sp.con_coef[k][f] : w_d * sp.load_coef[k][f];
        const bool loaded = !constrained && sp.load_coef[k][f] != 0.0;
        if ((coef != 0.0 && c.fn_g) || loaded) {
          double g, gt;
          if (TAB) {
            const u32 z = lane_zero();
            g = fn_tab_get(ft, f, 0, z);
            if (coef != 0.0 && c.fn_g) for (int kk = 0; kk < kMaxFnParams; ++kk) gp[kk] = fn_tab_get(ft, f, 2 + kk, z);
          } else eval_time_fn(c.fns[(u32)m * DFX_MAX_FNS + f], t_i, g, gt, gp);
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
    if (!REBUILD && !local_only) kq_in = h * (ac.cur[c.s] * lq + sqc);      // (zero on constrained DOFs: their lambda and Ybar are)
    if constexpr (DENSE) { if (!REBUILD && !local_only) kq_in += e_own_q; }
    if (!constrained) {
      ybq = -hw;
      ybv = kq_in - damp * w_d;
      stg_m<double>(WT, bmm, o_dof, bm_old - w_d * a_i);
      if (blk_c_) stg_m<double>(WT, bcm, o_dof, bc_old - w_d * v_i);
    }
    if (!REBUILD) stg_m<double2>(WT, YBm + (size_t)((u32)i * nd6), o_b6, make_double2(ybq, ybv));
    else { stg_m<double>(WT, YBm + (size_t)i * nd6, o_b6, ybq); stg_m<double>(WT, YBm + (size_t)i * nd6, o_b6 + 24, ybv); }
    if (!local_only) {
      double kq = 0.0, kv;       // Kbar of the next stage to run (records build: its Kbar_q is recomputed there)
      if (i > 0) {
        if (REBUILD) kq = h * (ac.col[c.s] * lq + ac.col[i] * ybq + sq);
        kv = h * (ac.col[c.s] * lv + ac.col[i] * ybv + sv);
      } else {
        lq += ybq + sq;
        lv += ybv + sv;
        if constexpr (DENSE) { lq += gs_q; lv += gs_v; }
        const bool first = !DENSE && (sg.j0 + j) == 0;
        if (first && c.G && !constrained) {
          const double* G = c.G + ((size_t)sg.interval * c.batch + m) * (size_t)nd6;
          lq += G[b * 6 + k]; lv += G[b * 6 + 3 + k];
        }
        if (constrained) { lq = 0.0; lv = 0.0; }
        if (!REBUILD) stg_m<double2>(WT, LAMm, o_b6, make_double2(lq, lv));
        else { stg_m<double>(WT, LAMm, o_b6, lq); stg_m<double>(WT, LAMm, o_b6 + 24, lv); }
        if (REBUILD) kq = h_before * ac.col[c.s] * lq;
        kv = h_before * ac.col[c.s] * lv;
      }
      if constexpr (DENSE) { kq += e_nxt_q; kv += e_nxt_v; }
      if (REBUILD) stg_m<double>(WT, c.KQ + (size_t)(((u32)m * 2 + (u32)(win ^ 1)) * nd), o_dof, kq);
      stg_m<double>(WT, (REBUILD ? c.W : DFX_LATE(double, W)) + (size_t)(((u32)m * 2 + (u32)(win ^ 1)) * nd), o_dof, constrained ? 0.0 : kv * invm);
    }
  }
  if (REBUILD && rb > 0) {
    if (i > 0) rebuild_record(c, m, b, k, rc, rb, n, h, t_n);
    else if (n > 0) rebuild_record(c, m, b, k, rc, rb, n - 1, h_before, c.t_steps ? steps_of(c, m)[n - 1] : t_n - h_before);
  }
}

template <int MODEL, int CONTACT, int BOND_GRADS, int REBUILD, int NPB = 4, int TAB = 0, int OVF = 0, int WT = 0, int ISTAGE = -1>
__global__ __launch_bounds__(kThreads) DFX_ADJ_OCC void k_adj_stage(DevCtx c, AdjCoef ac, int i, int j, int in_buf, int wbuf_static,
                                                        int local_only, StageCoef rc, int rb) {
  // the records build fetches its epilogue pointers from the kernel-argument segment at offsetof(DevCtx, field) (late_arg)
  static_assert(kernel_takes_devctx_first(&k_adj_stage<MODEL, CONTACT, BOND_GRADS, REBUILD, NPB, TAB, OVF, WT, ISTAGE>), "DevCtx must stay the first kernel argument");
  adj_stage_body<MODEL, CONTACT, BOND_GRADS, REBUILD, NPB, TAB, OVF, WT, ISTAGE>(c, ac, i, j, in_buf, wbuf_static, local_only, rc, rb);
}
// The stage-checkpoint build (REBUILD, no per-ligament gradients) sits at 127-131 VGPRs depending on unrelated edits: its own entry
// point, so that its occupancy can be pinned (DFX_ADJ_RB_OCC) without touching the others.
template <int MODEL, int CONTACT>
__global__ __launch_bounds__(kThreads) DFX_ADJ_RB_OCC void k_adj_stage_rb(DevCtx c, AdjCoef ac, int i, int j, int in_buf, int wbuf_static,
                                                           int local_only, StageCoef rc, int rb) {
  adj_stage_body<MODEL, CONTACT, 0, 1>(c, ac, i, j, in_buf, wbuf_static, local_only, rc, rb);
}

// start of the reverse sweep: lambda_N = G_last; kbar_{s-1} of the last step into buffer `buf`
//   lam_par: which of the two lambda buffers (the pair launches double-buffer lambda by step parity; 0 otherwise)
__global__ __launch_bounds__(kThreads) void k_adj_begin(DevCtx c, double h_last, double b_last, int buf, int lam_par, long long n_total) {
  const int m = blockIdx.y + c.m0;
  if (c.t_steps && n_total > 0) h_last = steps_of(c, m)[n_total] - steps_of(c, m)[n_total - 1];
  const int tid = blockIdx.x * kThreads + threadIdx.x;
  if (tid >= c.n_slots) return;
  const int b = tid >> 2, d = tid & 3;
  if (d == 3) return;
  const size_t nd = (size_t)c.n_blocks * 3, nd6 = (size_t)c.n_blocks * 6;
  const int sidx = c.block_special[b];
  const bool con = sidx >= 0 && ((c.special[sidx].con_mask >> d) & 1);
  const double* G = c.G + ((size_t)(c.n_timepoints - 1) * c.batch + m) * nd6;
  const double lq = con ? 0.0 : G[b * 6 + d], lv = con ? 0.0 : G[b * 6 + 3 + d];
  double* LAMm = c.LAM + ((size_t)lam_par * c.batch + m) * nd6;
  LAMm[b * 6 + (c.lam_pairs ? 2 * d : d)] = lq;
  LAMm[b * 6 + (c.lam_pairs ? 2 * d + 1 : 3 + d)] = lv;
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

// gradient accumulators -> the caller's layouts (dfx_grads), so that what crosses PCIe is final:
//   out_r   (batch, n_blocks, npb, 2)  node-vector gradients without the padding slot (kagome)
//   out_phi (batch, n_bonds, 2)        void-angle gradients: every ligament end holds one of the two
//   out_lam (batch, 2, n_blocks, 3)    state0 cotangent, (q | v) planes
__global__ __launch_bounds__(kThreads) void k_pack_grads(DevCtx c, const int32_t* slot_bond, int npb, int n_bonds, double* out_r,
                                                         double* out_phi, double* out_lam) {
  const int m = blockIdx.y;
  const int slot = blockIdx.x * kThreads + threadIdx.x;
  if (slot >= c.n_slots) return;
  const int b = slot >> 2, k = slot & 3;
  const size_t ms = (size_t)m * c.n_slots;
  if (out_r && k < npb)
    *reinterpret_cast<double2*>(out_r + (((size_t)m * c.n_blocks + b) * npb + k) * 2) = *reinterpret_cast<const double2*>(c.g_r + (ms + slot) * 2);
  if (out_phi) {
    const int info = c.slot_info[slot];
    if (info >= 0) out_phi[((size_t)m * n_bonds + slot_bond[slot]) * 2 + (info & 1)] = c.g_phi[ms + slot];
  }
  if (out_lam && k < 3) {
    const size_t nd = (size_t)c.n_blocks * 3, o = (size_t)m * c.n_blocks * 6;
    out_lam[o + (size_t)b * 3 + k] = c.LAM[o + (size_t)b * 6 + (c.lam_pairs ? 2 * k : k)];
    out_lam[o + nd + (size_t)b * 3 + k] = c.LAM[o + (size_t)b * 6 + (c.lam_pairs ? 2 * k + 1 : 3 + k)];
  }
}

// ---- post-processing of a solve on the device-resident history (problems/quads_focusing.py:319-372) -------------------------
// Per-ligament strain energies (stretch / shear / bending: energy.py:522-534 strains, then 1/2 k (strain l0)^2) and per-block
// kinetic energy of EVERY output time: one launch, grid = (slots, T, members); the end-0 lane of a ligament writes its three
// entries, lanes 0 of a quad the block's kinetic energy.  Reads the (T, 2, n, 3) fields and the resident parameter images.
__global__ __launch_bounds__(kThreads) void k_response(DevCtx c, const double* fields, const int32_t* slot_bond, const int32_t* ovf_bond, int n_bonds,
                                                       double* e_stretch, double* e_shear, double* e_bend, double* e_kin) {
  const int m = blockIdx.z, kt = blockIdx.y;
  const int slot = blockIdx.x * kThreads + threadIdx.x;
  if (slot >= c.n_slots) return;
  const int b = slot >> 2, k = slot & 3;
  const size_t nd = (size_t)c.n_blocks * 3;
  const double* f = fields + ((size_t)m * c.n_timepoints + kt) * nd * 2;
  if (e_kin && k == 0) {
    const double* im = c.inv_m + (size_t)m * nd + (size_t)b * 3;
    const double* v = f + nd + (size_t)b * 3;
    e_kin[((size_t)m * c.n_timepoints + kt) * c.n_blocks + b] = 0.5 * (v[0] * v[0] / im[0] + v[1] * v[1] / im[1] + v[2] * v[2] / im[2]);
  }
  if (!(e_stretch || e_shear || e_bend)) return;
  const MemberBases B = member_bases(c, m);
  // one lane per ligament: the lane of its end-0 node -- that node's first ligament, then its extra ones (general bond lists)
  const int n_extra = c.ovf_ptr ? c.ovf_ptr[slot + 1] - c.ovf_ptr[slot] : 0;
  for (int which = 0; which <= n_extra; ++which) {
    const int x = which ? c.ovf_ptr[slot] + which - 1 : -1;
    const int info = which ? c.ovf_info[x] : c.slot_info[slot];
    if (info < 0 || (info & 1)) continue;
    const int ps = info >> 1, pb = ps >> 2;
    BlockRec<double> o, p;
    o.x = f[(size_t)b * 3]; o.y = f[(size_t)b * 3 + 1]; o.th = f[(size_t)b * 3 + 2];
    p.x = f[(size_t)pb * 3]; p.y = f[(size_t)pb * 3 + 1]; p.th = f[(size_t)pb * 3 + 2];
    fast_sincos(0.5 * o.th, &o.sh, &o.ch);
    fast_sincos(0.5 * p.th, &p.sh, &p.ch);
    const double2 ro = ldg<double2>(B.p_r, (u32)slot * 16), rp = ldg<double2>(B.p_r, (u32)ps * 16);
    double lx, ly, l0, il0, ks, ksh, kr;
    if (which) {
      const double* op = B.ovf_p + (size_t)x * kOvfParams;
      lx = op[0]; ly = op[1]; l0 = sqrt(lx * lx + ly * ly); il0 = 1.0 / l0; ks = op[2]; ksh = op[3]; kr = op[4];
    } else {
      if (c.l_dict_on) {
        const int li = B.p_lidx[slot];
        lx = B.l_dict[li * 4]; ly = B.l_dict[li * 4 + 1]; l0 = B.l_dict[li * 4 + 2]; il0 = B.l_dict[li * 4 + 3];
      } else {
        lx = B.p_l[(size_t)slot * 2]; ly = B.p_l[(size_t)slot * 2 + 1]; l0 = sqrt(lx * lx + ly * ly); il0 = 1.0 / l0;
      }
      if (c.k_uniform) { ks = B.cst[3]; ksh = B.cst[4]; kr = B.cst[5]; }
      else { ks = B.p_k[(size_t)slot * 4]; ksh = B.p_k[(size_t)slot * 4 + 1]; kr = B.p_k[(size_t)slot * 4 + 2]; }
    }
    BondGrad<double> g;
    bond_grad<kNonlinear, double>(o, p, ro.x, ro.y, rp.x, rp.y, lx, ly, l0, il0, ks, ksh, kr, -1.0, g);   // g.ks = (eps l0)^2 / 2, ...
    const size_t oi = ((size_t)m * c.n_timepoints + kt) * n_bonds + (which ? ovf_bond[x] : slot_bond[slot]);
    if (e_stretch) e_stretch[oi] = ks * g.ks;
    if (e_shear) e_shear[oi] = ksh * g.ksh;
    if (e_bend) e_bend[oi] = kr * g.kr;
  }
}

// kinetic-energy objective: G <- m v on target blocks; per-member objective by one workgroup
__global__ __launch_bounds__(kThreads) void k_kinetic(DevCtx c, const double* fields, const int32_t* target, int n_target,
                                                      double* G, double* objective, double* objective_host) {
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
  if (threadIdx.x == 0 && objective_host) objective_host[m] = red[0];     // pinned host memory: no copy engine hop inside the stream
}

// Everything a reverse sweep clears or sets before its first launch, in ONE launch instead of a dozen fills and small copies queued one
// behind the other between the forward pass and the sweep (~85 us of a 5 ms job, profiles/r04_host_overhead.txt): up to kPreludeZero
// arrays of doubles zeroed (blockIdx.y picks the array), a run of ints set to one value (the groups' segment cursors), a few ints
// copied from the kernel arguments (the target blocks).
constexpr int kPreludeZero = 12, kPreludeInts = 32;
struct PreludeJob {
  double* zp[kPreludeZero];
  unsigned long long zn[kPreludeZero];
  int n_zero;
  int fill_val, fill_n, copy_n;
  int* fill_dst;
  int* copy_dst;
  int copy_val[kPreludeInts];
};
__global__ __launch_bounds__(256) void k_prelude(PreludeJob J) {
  const int job = blockIdx.y;
  if (job < J.n_zero) {
    double* p = J.zp[job];
    const unsigned long long n = J.zn[job], n2 = n >> 1;
    double2* p2 = reinterpret_cast<double2*>(p);          // (device allocations: 256-byte aligned)
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n2; i += (unsigned long long)gridDim.x * 256)
      p2[i] = make_double2(0.0, 0.0);
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) p[n - 1] = 0.0;
  }
  if (job == 0 && blockIdx.x == 0) {
    for (int i = threadIdx.x; i < J.fill_n; i += 256) J.fill_dst[i] = J.fill_val;
    for (int i = threadIdx.x; i < J.copy_n; i += 256) J.copy_dst[i] = J.copy_val[i];
  }
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
