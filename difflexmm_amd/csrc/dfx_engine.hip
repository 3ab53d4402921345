// dfx_engine.hip -- libdfx: the MI355X (gfx950) engine behind include/dfx.h.
//
// Execution model
//   * one lane per (block, node slot): 4 lanes = one rigid unit, 16 units per 64-wide wavefront,
//     256-thread workgroups, grid = (ceil(4*n_blocks/256), batch members).  Every ligament is evaluated
//     by both of its end lanes ("gather form"); the 4 slot contributions of a unit are summed with
//     quad shuffles; lanes 0..2 of the quad then own DOF x, y, theta for the integrator epilogue.
//     No atomics anywhere: results are bit-reproducible.
//   * one kernel launch per Runge-Kutta stage (the neighbour exchange of an explicit stage is a grid-wide
//     dependency; a kernel boundary is the cheapest grid barrier on this chip).  The
//     epilogue of stage i already assembles the stage record of stage i+1, so a launch only gathers
//     finished 64-byte block records (x y th cos(th/2) sin(th/2) vx vy vth).
//   * the time loop is replayed from hipGraphs (one graph = one segment of <= kMaxGraphSteps steps);
//     everything that changes between replays (time, step size, checkpoint slot) is read from a small
//     segment table in device memory, advanced by a 1-thread tick kernel at the head of each graph.
//   * the reverse sweep re-reads the checkpointed trajectory (one 64-B record per unit per step, kept in
//     HBM), recomputes the stage records of a step and runs one Dual-number kernel per stage.
//
// The per-lane physics is shared with the CPU port through dfx_physics.h / dfx_stage.h.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "dfx_stage.h"

using namespace dfx;

namespace {

constexpr int kThreads = 256;
constexpr int kMaxGraphSteps = 256;

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

// Everything a kernel needs; passed by value (kernarg).
struct DevCtx {
  int n_blocks, n_slots, n_fns, batch, s;
  long long traj_stride;  // elements between members in traj
  // static tables
  const int32_t* slot_info;
  const int32_t* block_special;
  const dfx_special* special;
  // per-member parameter images (member stride in elements)
  const double* slot_p;
  const double* inv_m;
  const double* damping;
  const double* contact_p;
  const TimeFn* fns;
  // time bookkeeping
  const Seg* segs;
  const int* seg_idx;
  // state
  double* traj;        // batch * (N+1) * n_blocks*kRec  (keep_trajectory) or null
  double* Ypp;         // batch * 2 * n_blocks*kRec      ping-pong step states when traj == null
  double* Sbuf;        // batch * (s+1) * n_blocks*kRec  stage records
  double* A;           // batch * s * n_blocks*3
  // reverse
  double* YB;          // batch * s * n_blocks*6
  double* LAM;         // batch * n_blocks*6
  double* W;           // batch * 2 * n_blocks*3
  double* KQ;          // batch * 2 * n_blocks*3
  const double* G;     // T * batch * n_blocks*6 (time-major, so graph kernels do not depend on T)
  int n_timepoints;
  double* slot_g;      // batch * n_slots*kSlotGrads
  double* blk_g;       // batch * n_blocks*6
  double* fn_g;        // batch * n_special*MAX_FNS*FN_PARAMS
  int n_special;
  Tableau tab;
};

__device__ __forceinline__ Tables dev_tables(const DevCtx& c, int m) {
  Tables tb;
  tb.n_blocks = c.n_blocks; tb.n_fns = c.n_fns; tb.model = 0; tb.contact = 0;
  tb.slot_info = c.slot_info;
  tb.block_special = c.block_special;
  tb.special = c.special;
  tb.slot_p = c.slot_p + (size_t)m * c.n_slots * kSlotParams;
  tb.inv_m = c.inv_m + (size_t)m * c.n_blocks * 3;
  tb.damping = c.damping + (size_t)m * c.n_blocks * 3;
  tb.contact_p = c.contact_p + (size_t)m * 3;
  tb.fns = c.fns + (size_t)m * DFX_MAX_FNS;
  return tb;
}

__device__ __forceinline__ double quad_sum(double v) {
  v += __shfl_xor(v, 1);
  v += __shfl_xor(v, 2);
  return v;
}

__device__ __forceinline__ double* step_state(const DevCtx& c, int m, long long n) {
  const size_t rec = (size_t)c.n_blocks * kRec;
  if (c.traj) return c.traj + (size_t)m * c.traj_stride + (size_t)n * rec;
  return c.Ypp + ((size_t)m * 2 + (n & 1)) * rec;
}

__global__ void k_tick(int* seg_idx, int delta) { *seg_idx += delta; }

// records of the initial state + row 0 of fields
__global__ __launch_bounds__(kThreads) void k_init(DevCtx c, const double* state0, double t0, double* rec_out /* batch stride n_blocks*kRec */) {
  int m = blockIdx.y;
  int tid = blockIdx.x * kThreads + threadIdx.x;
  if (tid >= c.n_slots) return;
  int b = tid >> 2, d = tid & 3;
  if (d == 3) return;
  Tables tb = dev_tables(c, m);
  init_dof(tb, state0 + (size_t)m * c.n_blocks * 6, t0, rec_out + (size_t)m * c.n_blocks * kRec, b, d);
}

// fields[m, k] <- record (disp, vel)
__global__ __launch_bounds__(kThreads) void k_snapshot(DevCtx c, const double* rec, size_t rec_member_stride, double* fields, int k) {
  int m = blockIdx.y;
  int tid = blockIdx.x * kThreads + threadIdx.x;
  if (tid >= c.n_blocks * 3) return;
  int b = tid / 3, d = tid % 3;
  const double* r = rec + (size_t)m * rec_member_stride + (size_t)b * kRec;
  double* f = fields + ((size_t)m * c.n_timepoints + k) * c.n_blocks * 6;
  f[tid] = r[d];
  f[(size_t)c.n_blocks * 3 + tid] = r[5 + d];
}

// ---- forward stage ---------------------------------------------------------------------------
// mode 0: regular time stepping (stage records ping-pong in Sbuf[0..1])
// mode 1: recompute for the reverse sweep (stage records kept in Sbuf[1..s-1], no step-end write)
// mode 2: test hook (single evaluation at stage 0, records in Sbuf[0], nothing written but A)
template <int MODEL, int CONTACT>
__global__ __launch_bounds__(kThreads) void k_fwd_stage(DevCtx c, int i, int j, int mode) {
  const int m = blockIdx.y;
  const int slot = blockIdx.x * kThreads + threadIdx.x;
  if (slot >= c.n_slots) return;
  const int b = slot >> 2, k = slot & 3;
  Tables tb = dev_tables(c, m);
  const size_t rec = (size_t)c.n_blocks * kRec;
  const Seg sg = c.segs[*c.seg_idx];
  const long long n = sg.base_step + j;
  const double t = sg.t_interval + (sg.j0 + j) * sg.h;
  double* Sm = c.Sbuf + (size_t)m * (c.s + 1) * rec;
  FwdStage st;
  st.i = i;
  st.h = sg.h;
  st.t_i = t + c.tab.c[i] * sg.h;
  st.t_next = t + c.tab.c[i + 1] * sg.h;
  st.A = c.A + (size_t)m * c.s * c.n_blocks * 3;
  if (mode == 2) {
    st.Y = Sm; st.S_in = Sm; st.S_out = nullptr;
  } else {
    st.Y = step_state(c, m, n);
    if (mode == 0) {
      st.S_in = i == 0 ? st.Y : Sm + (size_t)(i & 1) * rec;
      st.S_out = i == c.s - 1 ? step_state(c, m, n + 1) : Sm + (size_t)((i + 1) & 1) * rec;
    } else {
      st.S_in = i == 0 ? st.Y : Sm + (size_t)i * rec;
      st.S_out = i == c.s - 1 ? nullptr : Sm + (size_t)(i + 1) * rec;
    }
  }
  double fx, fy, fth;
  fwd_slot<MODEL, CONTACT>(tb, st.S_in, slot, fx, fy, fth, nullptr);
  fx = quad_sum(fx);
  fy = quad_sum(fy);
  fth = quad_sum(fth);
  if (k < 3) fwd_dof(tb, c.tab, st, b, k, k == 0 ? fx : (k == 1 ? fy : fth));
}

template <int MODEL, int CONTACT>
__global__ __launch_bounds__(kThreads) void k_energy(DevCtx c, double* e_slot) {
  const int m = blockIdx.y;
  const int slot = blockIdx.x * kThreads + threadIdx.x;
  if (slot >= c.n_slots) return;
  Tables tb = dev_tables(c, m);
  const size_t rec = (size_t)c.n_blocks * kRec;
  double fx, fy, fth, e;
  fwd_slot<MODEL, CONTACT>(tb, c.Sbuf + (size_t)m * (c.s + 1) * rec, slot, fx, fy, fth, &e);
  e_slot[(size_t)m * c.n_slots + slot] = e;
}

// ---- reverse stage ---------------------------------------------------------------------------
template <int MODEL, int CONTACT>
__global__ __launch_bounds__(kThreads) void k_adj_stage(DevCtx c, int i, int j, int local_only) {
  const int m = blockIdx.y;
  const int slot = blockIdx.x * kThreads + threadIdx.x;
  if (slot >= c.n_slots) return;
  const int b = slot >> 2, k = slot & 3;
  Tables tb = dev_tables(c, m);
  const size_t rec = (size_t)c.n_blocks * kRec;
  const size_t nd = (size_t)c.n_blocks * 3;
  const Seg sg = c.segs[*c.seg_idx];
  const long long n = sg.base_step + j;
  const double t = sg.t_interval + (sg.j0 + j) * sg.h;
  double* Sm = c.Sbuf + (size_t)m * (c.s + 1) * rec;
  // ping-pong parity of the (w, kbar_q) buffers: ordinal of this launch in the reverse sweep
  // forward ordinal of this stage; the reverse sweep visits ordinals in decreasing order, so parity alternates
  const int in = local_only ? 0 : (int)((n * c.s + i) & 1);
  AdjStage st;
  st.i = i;
  st.local_only = local_only;
  st.S = (i == 0 && !local_only) ? step_state(c, m, n) : Sm + (size_t)i * rec;
  st.A = c.A + (size_t)m * c.s * nd;
  st.W = c.W + ((size_t)m * 2 + in) * nd;
  st.KQ = c.KQ + ((size_t)m * 2 + in) * nd;
  st.W_out = c.W + ((size_t)m * 2 + (in ^ 1)) * nd;
  st.KQ_out = c.KQ + ((size_t)m * 2 + (in ^ 1)) * nd;
  st.YB = c.YB + (size_t)m * c.s * c.n_blocks * 6;
  st.LAM = c.LAM + (size_t)m * c.n_blocks * 6;
  const bool first_of_interval = (sg.j0 + j) == 0;
  st.G = (i == 0 && first_of_interval && c.G) ? c.G + ((size_t)sg.interval * c.batch + m) * c.n_blocks * 6 : nullptr;
  st.t_i = t + c.tab.c[i] * sg.h;
  st.h = sg.h;
  st.h_prev = first_of_interval ? sg.h_prev : sg.h;
  GradAcc acc;
  acc.slot_g = c.slot_g ? c.slot_g + (size_t)m * c.n_slots * kSlotGrads : nullptr;
  acc.blk_g = c.blk_g ? c.blk_g + (size_t)m * c.n_blocks * 6 : nullptr;
  acc.fn_g = c.fn_g ? c.fn_g + (size_t)m * c.n_special * DFX_MAX_FNS * DFX_FN_PARAMS : nullptr;
  double hx, hy, hth;
  adj_slot<MODEL, CONTACT>(tb, st.S, st.W, slot, acc, hx, hy, hth);
  hx = quad_sum(hx);
  hy = quad_sum(hy);
  hth = quad_sum(hth);
  if (k < 3) adj_dof(tb, c.tab, st, acc, b, k, k == 0 ? hx : (k == 1 ? hy : hth));
}

__global__ __launch_bounds__(kThreads) void k_adj_begin(DevCtx c, double h_last, int buf) {
  const int m = blockIdx.y;
  const int tid = blockIdx.x * kThreads + threadIdx.x;
  if (tid >= c.n_slots) return;
  const int b = tid >> 2, d = tid & 3;
  if (d == 3) return;
  Tables tb = dev_tables(c, m);
  const size_t nd = (size_t)c.n_blocks * 3;
  adj_begin_dof(tb, c.tab, c.G + ((size_t)(c.n_timepoints - 1) * c.batch + m) * c.n_blocks * 6, h_last,
                c.LAM + (size_t)m * c.n_blocks * 6, c.W + ((size_t)m * 2 + buf) * nd, c.KQ + ((size_t)m * 2 + buf) * nd, b, d);
}

// test hook: w = lam_v / m, kbar_q = lam_q  into buffer 0
__global__ __launch_bounds__(kThreads) void k_seed_vjp(DevCtx c, const double* lam) {
  const int m = blockIdx.y;
  const int tid = blockIdx.x * kThreads + threadIdx.x;
  if (tid >= c.n_blocks * 3) return;
  const int b = tid / 3, d = tid % 3;
  const size_t nd = (size_t)c.n_blocks * 3;
  int sidx = c.block_special[b];
  bool con = sidx >= 0 && ((c.special[sidx].con_mask >> d) & 1);
  const double* l = lam + (size_t)m * c.n_blocks * 6;
  c.W[(size_t)m * 2 * nd + tid] = con ? 0.0 : l[nd + tid] * c.inv_m[(size_t)m * nd + tid];
  c.KQ[(size_t)m * 2 * nd + tid] = l[tid];
}

// fields_bar (batch, T, 2, n_blocks, 3) -> G (batch, T, n_blocks, 6)
__global__ __launch_bounds__(kThreads) void k_pack_G(DevCtx c, const double* fields_bar, double* G) {
  const size_t total = (size_t)c.batch * c.n_timepoints * c.n_blocks * 3;
  size_t tid = (size_t)blockIdx.x * kThreads + threadIdx.x;
  if (tid >= total) return;
  size_t mk = tid / ((size_t)c.n_blocks * 3);
  int r = (int)(tid % ((size_t)c.n_blocks * 3));
  int b = r / 3, d = r % 3;
  const double* f = fields_bar + mk * c.n_blocks * 6;
  const size_t m_ = mk / c.n_timepoints, k_ = mk % c.n_timepoints;
  double* g = G + (k_ * c.batch + m_) * c.n_blocks * 6;   // G is time-major: (T, batch, n_blocks, 6)
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
    int k = (int)(idx / per_t), r = (int)(idx % per_t);
    int b = target[r / 3], d = r % 3;
    double v = fields[((size_t)m * c.n_timepoints + k) * c.n_blocks * 6 + (size_t)c.n_blocks * 3 + b * 3 + d];
    double mass = 1.0 / c.inv_m[(size_t)m * c.n_blocks * 3 + b * 3 + d];
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

// explicit d(objective)/d(inertia) = sum_t v^2/2 on target DOFs, added to blk_g
__global__ void k_kinetic_mass_grad(DevCtx c, const double* fields, const int32_t* target, int n_target) {
  const int m = blockIdx.y;
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_target * 3) return;
  int b = target[r / 3], d = r % 3;
  double acc = 0.0;
  for (int k = 0; k < c.n_timepoints; ++k) {
    double v = fields[((size_t)m * c.n_timepoints + k) * c.n_blocks * 6 + (size_t)c.n_blocks * 3 + b * 3 + d];
    acc += 0.5 * v * v;
  }
  c.blk_g[((size_t)m * c.n_blocks + b) * 6 + d] += acc;
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
    if (e == hipSuccess) n = count;
    return e;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
};

struct dfx_handle {
  Plan pl;
  PackedParams pp;
  std::string err;
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool have_params = false, have_traj = false, have_fields = false;
  bool use_graph = true;
  // static
  DevBuf<int32_t> d_slot_info, d_block_special;
  DevBuf<dfx_special> d_special;
  // params
  DevBuf<double> d_slot_p, d_inv_m, d_damping, d_contact;
  DevBuf<TimeFn> d_fns;
  // time
  DevBuf<Seg> d_segs;
  DevBuf<int> d_seg_idx;
  std::vector<Seg> segs;
  // state
  DevBuf<double> d_traj, d_Ypp, d_Sbuf, d_A, d_state0, d_fields;
  DevBuf<double> d_YB, d_LAM, d_W, d_KQ, d_G, d_slot_g, d_blk_g, d_fn_g, d_tmp, d_obj;
  DevBuf<int32_t> d_target;
  std::vector<double> ts;
  int spi = 0;
  long long n_total = 0;
  // graphs: key = (n_steps, kind) ; invalidated when buffers move
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
  c.traj_stride = h->pl.batch ? (long long)(h->d_traj.n / h->pl.batch) : 0;
  c.slot_info = h->d_slot_info.p; c.block_special = h->d_block_special.p; c.special = h->d_special.p;
  c.slot_p = h->d_slot_p.p; c.inv_m = h->d_inv_m.p; c.damping = h->d_damping.p; c.contact_p = h->d_contact.p;
  c.fns = h->d_fns.p;
  c.segs = h->d_segs.p; c.seg_idx = h->d_seg_idx.p;
  c.traj = h->have_traj ? h->d_traj.p : nullptr;
  c.Ypp = h->d_Ypp.p; c.Sbuf = h->d_Sbuf.p; c.A = h->d_A.p;
  c.YB = h->d_YB.p; c.LAM = h->d_LAM.p; c.W = h->d_W.p; c.KQ = h->d_KQ.p; c.G = h->d_G.p;
  c.n_timepoints = (int)h->ts.size();
  c.slot_g = h->d_slot_g.p; c.blk_g = h->d_blk_g.p; c.fn_g = h->d_fn_g.p;
  c.n_special = pl.n_special;
  c.tab = pl.tab;
  return c;
}

static dim3 slot_grid(const dfx_handle* h) { return dim3((h->pl.n_slots + kThreads - 1) / kThreads, h->pl.batch); }

template <int MODEL, int CONTACT>
static void launch_fwd_t(dfx_handle* h, const DevCtx& c, int i, int j, int mode) {
  hipLaunchKernelGGL((k_fwd_stage<MODEL, CONTACT>), slot_grid(h), dim3(kThreads), 0, h->stream, c, i, j, mode);
}
static void launch_fwd(dfx_handle* h, const DevCtx& c, int i, int j, int mode) {
  const Plan& pl = h->pl;
  if (pl.model == kNonlinear) { if (pl.contact) launch_fwd_t<kNonlinear, 1>(h, c, i, j, mode); else launch_fwd_t<kNonlinear, 0>(h, c, i, j, mode); }
  else { if (pl.contact) launch_fwd_t<kLinearized, 1>(h, c, i, j, mode); else launch_fwd_t<kLinearized, 0>(h, c, i, j, mode); }
  h->launches++;
}
template <int MODEL, int CONTACT>
static void launch_adj_t(dfx_handle* h, const DevCtx& c, int i, int j, int local_only) {
  hipLaunchKernelGGL((k_adj_stage<MODEL, CONTACT>), slot_grid(h), dim3(kThreads), 0, h->stream, c, i, j, local_only);
}
static void launch_adj(dfx_handle* h, const DevCtx& c, int i, int j, int local_only) {
  const Plan& pl = h->pl;
  if (pl.model == kNonlinear) { if (pl.contact) launch_adj_t<kNonlinear, 1>(h, c, i, j, local_only); else launch_adj_t<kNonlinear, 0>(h, c, i, j, local_only); }
  else { if (pl.contact) launch_adj_t<kLinearized, 1>(h, c, i, j, local_only); else launch_adj_t<kLinearized, 0>(h, c, i, j, local_only); }
  h->launches++;
}

// enqueue one segment (kind 0: forward steps; kind 1: reverse steps) on h->stream
static void enqueue_segment(dfx_handle* h, const DevCtx& c, int n_steps, int kind) {
  const int s = h->pl.tab.s;
  if (kind == 0) {
    hipLaunchKernelGGL(k_tick, dim3(1), dim3(1), 0, h->stream, h->d_seg_idx.p, 1);
    for (int j = 0; j < n_steps; ++j)
      for (int i = 0; i < s; ++i) launch_fwd(h, c, i, j, 0);
  } else {
    hipLaunchKernelGGL(k_tick, dim3(1), dim3(1), 0, h->stream, h->d_seg_idx.p, -1);
    for (int j = n_steps - 1; j >= 0; --j) {
      for (int i = 0; i < s; ++i) launch_fwd(h, c, i, j, 1);
      for (int i = s - 1; i >= 0; --i) launch_adj(h, c, i, j, 0);
    }
  }
  h->launches++;
}

static int run_segment(dfx_handle* h, const DevCtx& c, int n_steps, int kind) {
  if (!h->use_graph) { enqueue_segment(h, c, n_steps, kind); return 0; }
  DevCtx key_ctx = c;
  key_ctx.n_timepoints = 0;  // not read by the stage kernels
  if (!h->graph_ctx_valid || memcmp(&h->graph_ctx_snapshot, &key_ctx, sizeof(DevCtx)) != 0) {
    drop_graphs(h);
    h->graph_ctx_snapshot = key_ctx;
    h->graph_ctx_valid = true;
  }
  auto key = std::make_pair(n_steps, kind);
  auto it = h->graphs.find(key);
  const int s = h->pl.tab.s;
  const long long per = 1 + (long long)n_steps * s * (kind == 0 ? 1 : 2);
  if (it == h->graphs.end()) {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    long long before = h->launches;
    HIP_OK(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
    enqueue_segment(h, c, n_steps, kind);
    HIP_OK(hipStreamEndCapture(h->stream, &graph));
    h->launches = before;
    HIP_OK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    (void)hipGraphDestroy(graph);
    it = h->graphs.emplace(key, exec).first;
  }
  HIP_OK(hipGraphLaunch(it->second, h->stream));
  h->launches += per;
  return 0;
}

static void build_segments(dfx_handle* h) {
  h->segs.clear();
  const int Tn = (int)h->ts.size();
  for (int k = 0; k + 1 < Tn; ++k) {
    const double hh = (h->ts[k + 1] - h->ts[k]) / h->spi;
    const double hp = k > 0 ? (h->ts[k] - h->ts[k - 1]) / h->spi : 0.0;
    for (int j0 = 0; j0 < h->spi; j0 += kMaxGraphSteps) {
      Seg sg;
      sg.t_interval = h->ts[k]; sg.h = hh; sg.h_prev = hp;
      sg.base_step = (long long)k * h->spi + j0; sg.j0 = j0; sg.interval = k;
      sg.n_steps = std::min(kMaxGraphSteps, h->spi - j0); sg.pad = 0;
      h->segs.push_back(sg);
    }
  }
}

extern "C" {

int dfx_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char* dfx_version(void) { return "dfx-hip-gfx950 0.1.0"; }

const char* dfx_last_error(const dfx_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int dfx_create(const dfx_problem* problem, dfx_handle** out) {
  dfx_handle* h = new dfx_handle();
  auto fail = [&](int rc) { g_create_error = h->err; delete h; return rc; };
  if (build_plan(problem, h->pl, h->err)) return fail(1);
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0) { h->err = "no HIP device available (libdfx has no CPU fallback)"; return fail(2); }
  if (problem->device < 0 || problem->device >= ndev) { h->err = "device ordinal out of range"; return fail(1); }
  h->device = problem->device;
  if (hipSetDevice(h->device) != hipSuccess) { h->err = "hipSetDevice failed"; return fail(2); }
  if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) { h->err = "hipStreamCreate failed"; return fail(2); }
  (void)hipEventCreate(&h->ev0);
  (void)hipEventCreate(&h->ev1);
  const char* g = getenv("DFX_NO_GRAPH");
  h->use_graph = !(g && g[0] == '1');
  const Plan& pl = h->pl;
  bool ok = h->d_slot_info.ensure(pl.n_slots) == hipSuccess && h->d_block_special.ensure(pl.n_blocks) == hipSuccess &&
            h->d_special.ensure(std::max(1, pl.n_special)) == hipSuccess && h->d_seg_idx.ensure(1) == hipSuccess;
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
  h->d_slot_p.release(); h->d_inv_m.release(); h->d_damping.release(); h->d_contact.release(); h->d_fns.release();
  h->d_segs.release(); h->d_seg_idx.release();
  h->d_traj.release(); h->d_Ypp.release(); h->d_Sbuf.release(); h->d_A.release(); h->d_state0.release(); h->d_fields.release();
  h->d_YB.release(); h->d_LAM.release(); h->d_W.release(); h->d_KQ.release(); h->d_G.release();
  h->d_slot_g.release(); h->d_blk_g.release(); h->d_fn_g.release(); h->d_tmp.release(); h->d_obj.release(); h->d_target.release();
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return 0;
}

static int ensure_work_buffers(dfx_handle* h) {
  const Plan& pl = h->pl;
  const size_t B = pl.batch, nb = pl.n_blocks, s = pl.tab.s, rec = nb * kRec;
  HIP_OK(h->d_Ypp.ensure(B * 2 * rec));
  HIP_OK(h->d_Sbuf.ensure(B * (s + 1) * rec));
  HIP_OK(h->d_A.ensure(B * s * nb * 3));
  HIP_OK(h->d_state0.ensure(B * nb * 6));
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
  HIP_OK(h->d_slot_g.ensure(B * pl.n_slots * kSlotGrads));
  HIP_OK(h->d_blk_g.ensure(B * nb * 6));
  HIP_OK(h->d_fn_g.ensure(B * nsp * DFX_MAX_FNS * DFX_FN_PARAMS));
  return 0;
}

int dfx_set_params(dfx_handle* h, const dfx_params* params) {
  HIP_OK(hipSetDevice(h->device));
  if (pack_params(h->pl, params, h->pp, h->err)) return 1;
  const PackedParams& pp = h->pp;
  HIP_OK(h->d_slot_p.ensure(pp.slot.size()));
  HIP_OK(h->d_inv_m.ensure(pp.inv_m.size()));
  HIP_OK(h->d_damping.ensure(pp.damping.size()));
  HIP_OK(h->d_contact.ensure(pp.contact.size()));
  HIP_OK(h->d_fns.ensure(pp.fns.size()));
  HIP_OK(hipMemcpyAsync(h->d_slot_p.p, pp.slot.data(), sizeof(double) * pp.slot.size(), hipMemcpyHostToDevice, h->stream));
  HIP_OK(hipMemcpyAsync(h->d_inv_m.p, pp.inv_m.data(), sizeof(double) * pp.inv_m.size(), hipMemcpyHostToDevice, h->stream));
  HIP_OK(hipMemcpyAsync(h->d_damping.p, pp.damping.data(), sizeof(double) * pp.damping.size(), hipMemcpyHostToDevice, h->stream));
  HIP_OK(hipMemcpyAsync(h->d_contact.p, pp.contact.data(), sizeof(double) * pp.contact.size(), hipMemcpyHostToDevice, h->stream));
  HIP_OK(hipMemcpyAsync(h->d_fns.p, pp.fns.data(), sizeof(TimeFn) * pp.fns.size(), hipMemcpyHostToDevice, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  h->have_params = true;
  h->have_traj = false;
  h->have_fields = false;
  return 0;
}

int dfx_reserve(dfx_handle* h, int64_t max_steps, int32_t max_timepoints, int32_t keep_trajectory) {
  HIP_OK(hipSetDevice(h->device));
  const Plan& pl = h->pl;
  const size_t B = pl.batch, nb = pl.n_blocks, rec = nb * kRec;
  if (ensure_work_buffers(h)) return 2;
  if (ensure_adjoint_buffers(h)) return 2;
  HIP_OK(h->d_fields.ensure(B * (size_t)max_timepoints * nb * 6));
  HIP_OK(h->d_G.ensure(B * (size_t)max_timepoints * nb * 6));
  HIP_OK(h->d_tmp.ensure(B * (size_t)max_timepoints * nb * 6));
  HIP_OK(h->d_target.ensure(nb));
  HIP_OK(h->d_obj.ensure(B));
  HIP_OK(h->d_segs.ensure((size_t)max_timepoints * (1 + (size_t)(max_steps / std::max(1, max_timepoints - 1)) / kMaxGraphSteps + 1)));
  if (keep_trajectory) {
    hipError_t e = h->d_traj.ensure(B * (size_t)(max_steps + 1) * rec);
    if (e != hipSuccess) { h->err = "reserve: cannot allocate the trajectory checkpoint"; return 2; }
  }
  return 0;
}

int dfx_forward(dfx_handle* h, const double* state0, const double* timepoints, int32_t n_timepoints,
                int32_t steps_per_interval, int32_t keep_trajectory, double* fields, dfx_stats* stats) {
  HIP_OK(hipSetDevice(h->device));
  if (!h->have_params) { h->err = "forward: set_params first"; return 1; }
  if (n_timepoints < 1 || steps_per_interval < 1) { h->err = "forward: need >= 1 timepoint and >= 1 step per interval"; return 1; }
  const Plan& pl = h->pl;
  const size_t B = pl.batch, nb = pl.n_blocks, rec = nb * kRec;
  const int Tn = n_timepoints;
  h->ts.assign(timepoints, timepoints + Tn);
  h->spi = steps_per_interval;
  h->n_total = (long long)(Tn - 1) * steps_per_interval;
  if (ensure_work_buffers(h)) return 2;
  HIP_OK(h->d_fields.ensure(B * Tn * nb * 6));
  h->have_traj = false;
  if (keep_trajectory) {
    hipError_t e = h->d_traj.ensure(B * (size_t)(h->n_total + 1) * rec);
    if (e != hipSuccess) { h->err = "forward: cannot allocate the trajectory checkpoint (" + std::to_string(B * (h->n_total + 1) * rec * 8 >> 20) + " MiB)"; return 2; }
    h->have_traj = true;
  }
  build_segments(h);
  HIP_OK(h->d_segs.ensure(std::max<size_t>(1, h->segs.size())));
  if (!h->segs.empty())
    HIP_OK(hipMemcpyAsync(h->d_segs.p, h->segs.data(), sizeof(Seg) * h->segs.size(), hipMemcpyHostToDevice, h->stream));
  int minus1 = -1;
  HIP_OK(hipMemcpyAsync(h->d_seg_idx.p, &minus1, sizeof(int), hipMemcpyHostToDevice, h->stream));
  HIP_OK(hipMemcpyAsync(h->d_state0.p, state0, sizeof(double) * B * nb * 6, hipMemcpyHostToDevice, h->stream));
  DevCtx c = make_ctx(h);
  h->launches = 0;
  // initial records: step 0 state
  double* y0 = c.traj ? c.traj : c.Ypp;
  const size_t y_stride = c.traj ? (size_t)c.traj_stride : 2 * rec;
  // k_init writes with member stride n_blocks*kRec; write into Sbuf[0] then copy per member
  hipLaunchKernelGGL(k_init, slot_grid(h), dim3(kThreads), 0, h->stream, c, (const double*)h->d_state0.p, timepoints[0], h->d_Sbuf.p);
  for (size_t m = 0; m < B; ++m)
    HIP_OK(hipMemcpyAsync(y0 + m * y_stride, h->d_Sbuf.p + m * rec, sizeof(double) * rec, hipMemcpyDeviceToDevice, h->stream));
  dim3 g3((unsigned)((nb * 3 + kThreads - 1) / kThreads), (unsigned)B);
  hipLaunchKernelGGL(k_snapshot, g3, dim3(kThreads), 0, h->stream, c, (const double*)y0, y_stride, h->d_fields.p, 0);
  HIP_OK(hipEventRecord(h->ev0, h->stream));
  for (size_t si = 0; si < h->segs.size(); ++si) {
    const Seg& sg = h->segs[si];
    int rc = run_segment(h, c, sg.n_steps, 0);
    if (rc) return rc;
    if (sg.j0 + sg.n_steps == h->spi) {
      const long long n_end = sg.base_step + sg.n_steps;
      const double* yend = c.traj ? c.traj + (size_t)n_end * rec : c.Ypp + (size_t)(n_end & 1) * rec;
      hipLaunchKernelGGL(k_snapshot, g3, dim3(kThreads), 0, h->stream, c, yend, y_stride, h->d_fields.p, sg.interval + 1);
    }
  }
  HIP_OK(hipEventRecord(h->ev1, h->stream));
  if (fields) HIP_OK(hipMemcpyAsync(fields, h->d_fields.p, sizeof(double) * B * Tn * nb * 6, hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  HIP_OK(hipGetLastError());
  h->have_fields = true;
  if (stats) {
    memset(stats, 0, sizeof(*stats));
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, h->ev0, h->ev1);
    stats->steps = h->n_total;
    stats->rhs_evals = h->n_total * pl.tab.s;
    stats->launches = h->launches;
    stats->kernel_ms = ms;
    stats->stage_kernel_us = h->n_total ? 1e3 * ms / (double)(h->n_total * pl.tab.s) : 0.0;
  }
  return 0;
}

// reverse sweep with G already in h->d_G
static int run_adjoint(dfx_handle* h, dfx_grads* grads, dfx_stats* stats, bool kinetic, int n_target) {
  const Plan& pl = h->pl;
  const size_t B = pl.batch, nb = pl.n_blocks;
  const int Tn = (int)h->ts.size();
  const size_t nsp = std::max(1, pl.n_special);
  DevCtx c = make_ctx(h);
  h->launches = 0;
  HIP_OK(hipMemsetAsync(h->d_slot_g.p, 0, sizeof(double) * B * pl.n_slots * kSlotGrads, h->stream));
  HIP_OK(hipMemsetAsync(h->d_blk_g.p, 0, sizeof(double) * B * nb * 6, h->stream));
  HIP_OK(hipMemsetAsync(h->d_fn_g.p, 0, sizeof(double) * B * nsp * DFX_MAX_FNS * DFX_FN_PARAMS, h->stream));
  int nseg = (int)h->segs.size();
  HIP_OK(hipMemcpyAsync(h->d_seg_idx.p, &nseg, sizeof(int), hipMemcpyHostToDevice, h->stream));
  const double h_last = Tn > 1 ? (h->ts[Tn - 1] - h->ts[Tn - 2]) / h->spi : 0.0;
  HIP_OK(hipEventRecord(h->ev0, h->stream));
  hipLaunchKernelGGL(k_adj_begin, slot_grid(h), dim3(kThreads), 0, h->stream, c, h_last, (int)((h->n_total * pl.tab.s - 1) & 1));
  for (int si = nseg - 1; si >= 0; --si) {
    int rc = run_segment(h, c, h->segs[si].n_steps, 1);
    if (rc) return rc;
  }
  if (kinetic) {
    dim3 g((unsigned)((n_target * 3 + 63) / 64), (unsigned)B);
    hipLaunchKernelGGL(k_kinetic_mass_grad, g, dim3(64), 0, h->stream, c, (const double*)h->d_fields.p, (const int32_t*)h->d_target.p, n_target);
  }
  HIP_OK(hipEventRecord(h->ev1, h->stream));
  std::vector<double> slot_g(B * pl.n_slots * kSlotGrads), blk_g(B * nb * 6), fn_g(B * nsp * DFX_MAX_FNS * DFX_FN_PARAMS), lam(B * nb * 6);
  HIP_OK(hipMemcpyAsync(slot_g.data(), h->d_slot_g.p, sizeof(double) * slot_g.size(), hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipMemcpyAsync(blk_g.data(), h->d_blk_g.p, sizeof(double) * blk_g.size(), hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipMemcpyAsync(fn_g.data(), h->d_fn_g.p, sizeof(double) * fn_g.size(), hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipMemcpyAsync(lam.data(), h->d_LAM.p, sizeof(double) * lam.size(), hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  HIP_OK(hipGetLastError());
  if (grads) {
    unpack_grads(pl, slot_g, blk_g, fn_g, h->pp.inv_m, grads);
    if (grads->state0)
      for (size_t m = 0; m < B; ++m)
        for (size_t b = 0; b < nb; ++b)
          for (int d = 0; d < 3; ++d) {
            grads->state0[m * nb * 6 + b * 3 + d] = lam[m * nb * 6 + b * 6 + d];
            grads->state0[m * nb * 6 + nb * 3 + b * 3 + d] = lam[m * nb * 6 + b * 6 + 3 + d];
          }
  }
  if (stats) {
    memset(stats, 0, sizeof(*stats));
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, h->ev0, h->ev1);
    stats->steps = h->n_total;
    stats->rhs_evals = h->n_total * pl.tab.s;
    stats->launches = h->launches;
    stats->kernel_ms = ms;
    stats->stage_kernel_us = h->n_total ? 1e3 * ms / (double)(h->n_total * pl.tab.s * 2) : 0.0;
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
  size_t total = B * Tn * nb * 3;
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
  if (ensure_work_buffers(h)) return 2;
  if (ensure_adjoint_buffers(h)) return 2;
  h->have_traj = false;
  h->ts.assign(1, t);
  h->n_total = 1;
  Seg sg;
  sg.t_interval = t; sg.h = 0.0; sg.h_prev = 0.0; sg.base_step = 0; sg.j0 = 0; sg.interval = 0; sg.n_steps = 1; sg.pad = 0;
  HIP_OK(h->d_segs.ensure(1));
  HIP_OK(hipMemcpyAsync(h->d_segs.p, &sg, sizeof(Seg), hipMemcpyHostToDevice, h->stream));
  int zero = 0;
  HIP_OK(hipMemcpyAsync(h->d_seg_idx.p, &zero, sizeof(int), hipMemcpyHostToDevice, h->stream));
  HIP_OK(hipMemcpyAsync(h->d_state0.p, y, sizeof(double) * B * nb * 6, hipMemcpyHostToDevice, h->stream));
  DevCtx c = make_ctx(h);
  c.G = nullptr;
  // records into Sbuf[0] of every member: k_init uses member stride n_blocks*kRec, Sbuf uses (s+1)*that
  HIP_OK(h->d_tmp.ensure(B * nb * kRec));
  hipLaunchKernelGGL(k_init, slot_grid(h), dim3(kThreads), 0, h->stream, c, (const double*)h->d_state0.p, t, h->d_tmp.p);
  for (size_t m = 0; m < B; ++m)
    HIP_OK(hipMemcpyAsync(h->d_Sbuf.p + m * (pl.tab.s + 1) * nb * kRec, h->d_tmp.p + m * nb * kRec, sizeof(double) * nb * kRec,
                          hipMemcpyDeviceToDevice, h->stream));
  return 0;
}

int dfx_rhs(dfx_handle* h, const double* y, double t, double* dy) {
  HIP_OK(hipSetDevice(h->device));
  if (int rc = hook_prepare(h, y, t)) return rc;
  const Plan& pl = h->pl;
  const size_t B = pl.batch, nb = pl.n_blocks;
  DevCtx c = make_ctx(h);
  launch_fwd(h, c, 0, 0, 2);
  std::vector<double> A(B * pl.tab.s * nb * 3), S(B * (pl.tab.s + 1) * nb * kRec);
  HIP_OK(hipMemcpyAsync(A.data(), h->d_A.p, sizeof(double) * A.size(), hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipMemcpyAsync(S.data(), h->d_Sbuf.p, sizeof(double) * S.size(), hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  HIP_OK(hipGetLastError());
  for (size_t m = 0; m < B; ++m)
    for (size_t b = 0; b < nb; ++b)
      for (int d = 0; d < 3; ++d) {
        int sidx = pl.block_special[b];
        bool con = sidx >= 0 && ((pl.special[sidx].con_mask >> d) & 1);
        dy[m * nb * 6 + b * 3 + d] = con ? 0.0 : S[m * (pl.tab.s + 1) * nb * kRec + b * kRec + 5 + d];
        dy[m * nb * 6 + nb * 3 + b * 3 + d] = A[m * pl.tab.s * nb * 3 + b * 3 + d];
      }
  return 0;
}

int dfx_rhs_vjp(dfx_handle* h, const double* y, double t, const double* lam, double* y_bar, dfx_grads* grads) {
  HIP_OK(hipSetDevice(h->device));
  if (int rc = hook_prepare(h, y, t)) return rc;
  const Plan& pl = h->pl;
  const size_t B = pl.batch, nb = pl.n_blocks;
  const size_t nsp = std::max(1, pl.n_special);
  DevCtx c = make_ctx(h);
  c.G = nullptr;
  launch_fwd(h, c, 0, 0, 2);
  HIP_OK(h->d_G.ensure(B * nb * 6));
  HIP_OK(hipMemcpyAsync(h->d_G.p, lam, sizeof(double) * B * nb * 6, hipMemcpyHostToDevice, h->stream));
  HIP_OK(hipMemsetAsync(h->d_slot_g.p, 0, sizeof(double) * B * pl.n_slots * kSlotGrads, h->stream));
  HIP_OK(hipMemsetAsync(h->d_blk_g.p, 0, sizeof(double) * B * nb * 6, h->stream));
  HIP_OK(hipMemsetAsync(h->d_fn_g.p, 0, sizeof(double) * B * nsp * DFX_MAX_FNS * DFX_FN_PARAMS, h->stream));
  dim3 g3((unsigned)((nb * 3 + kThreads - 1) / kThreads), (unsigned)B);
  hipLaunchKernelGGL(k_seed_vjp, g3, dim3(kThreads), 0, h->stream, c, (const double*)h->d_G.p);
  launch_adj(h, c, 0, 0, 1);
  std::vector<double> YB(B * pl.tab.s * nb * 6), slot_g(B * pl.n_slots * kSlotGrads), blk_g(B * nb * 6), fn_g(B * nsp * DFX_MAX_FNS * DFX_FN_PARAMS);
  HIP_OK(hipMemcpyAsync(YB.data(), h->d_YB.p, sizeof(double) * YB.size(), hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipMemcpyAsync(slot_g.data(), h->d_slot_g.p, sizeof(double) * slot_g.size(), hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipMemcpyAsync(blk_g.data(), h->d_blk_g.p, sizeof(double) * blk_g.size(), hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipMemcpyAsync(fn_g.data(), h->d_fn_g.p, sizeof(double) * fn_g.size(), hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  HIP_OK(hipGetLastError());
  for (size_t m = 0; m < B; ++m)
    for (size_t b = 0; b < nb; ++b)
      for (int d = 0; d < 3; ++d) {
        y_bar[m * nb * 6 + b * 3 + d] = YB[m * pl.tab.s * nb * 6 + b * 6 + d];
        y_bar[m * nb * 6 + nb * 3 + b * 3 + d] = YB[m * pl.tab.s * nb * 6 + b * 6 + 3 + d];
      }
  if (grads) {
    dfx_grads g = *grads;
    g.state0 = nullptr;
    unpack_grads(pl, slot_g, blk_g, fn_g, h->pp.inv_m, &g);
  }
  return 0;
}

int dfx_energy(dfx_handle* h, const double* u, double* energy) {
  HIP_OK(hipSetDevice(h->device));
  const Plan& pl = h->pl;
  const size_t B = pl.batch, nb = pl.n_blocks;
  if (!h->have_params) { h->err = "energy: set_params first"; return 1; }
  if (ensure_work_buffers(h)) return 2;
  std::vector<double> S(B * (pl.tab.s + 1) * nb * kRec, 0.0);
  for (size_t m = 0; m < B; ++m)
    for (size_t b = 0; b < nb; ++b) {
      double* r = S.data() + m * (pl.tab.s + 1) * nb * kRec + b * kRec;
      for (int d = 0; d < 3; ++d) r[d] = u[m * nb * 3 + b * 3 + d];
      r[3] = cos(0.5 * r[2]); r[4] = sin(0.5 * r[2]);
    }
  HIP_OK(hipMemcpyAsync(h->d_Sbuf.p, S.data(), sizeof(double) * S.size(), hipMemcpyHostToDevice, h->stream));
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
