// dfx_engine.h -- what the translation units of libdfx's host side share: the handle, its buffers, and the functions that cross files.
//   engine_launch.hip    which kernel build a stage takes, segments as stage launches / hipGraphs / persistent launches, member groups
//   engine_forward.hip   fixed-grid forward solve, checkpoint levels, segments
//   engine_adaptive.hip  the reference's adaptive odeint semantics
//   engine_reverse.hip   reverse sweep, gradient collection, objectives
//   engine_dense.hip     reverse sweep of an adaptive solve that kept its accepted steps (dense-output discrete adjoint)
//   engine_abi.hip       create / destroy / set_params / reserve, test hooks, post-processing, downloads
// The kernels live in dfx_kernels.h (stage kernels: instantiated by engine_launch.hip, the reverse per-stage builds by stage_builds_adj.hip)
// and dfx_persist.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "dfx_kernels.h"
#include "dfx_persist_api.h"
#ifdef DFX_EXPERIMENTAL
#include "dfx_pair.h"
#include "dfx_tile.h"
#endif

#define HIP_OK(call)                                                                                   \
  do {                                                                                                 \
    hipError_t e_ = (call);                                                                            \
    if (e_ != hipSuccess) {                                                                            \
      h->err = std::string(#call) + ": " + hipGetErrorString(e_);                                      \
      return 2;                                                                                        \
    }                                                                                                  \
  } while (0)


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

// The trajectory checkpoint (and the stage accelerations of the "stages" level) is by far the largest allocation of a handle.  Handles
// whose solves never overlap in time -- the engines of a multi-input objective, evaluated one input after the other, forward + reverse
// each -- can share ONE (dfx_share_checkpoint): a third of the memory, a third of the allocation time, and room for a richer level.
// `writer` is the handle whose forward pass filled it last: a reverse sweep of any other handle refuses to run on it.
struct CheckpointPool {
  DevBuf<double> traj, AD;
  int users = 1;
  const void* writer = nullptr;
};

struct dfx_handle {
  CheckpointPool* ck = new CheckpointPool();
  Plan pl;
  std::vector<Group> groups;
  bool dual_chain = true;
  hipEvent_t ev_fork2 = nullptr;
  // flag_stage: one-word results (non-finite flag, touched flag) land in pinned memory; zero_phi: an all-zero void-angle gradient handed out when the sweep
  // never touched the accumulator
  PinnedBuf stage, obj_stage, flag_stage, zero_phi;
  hipEvent_t ev_fork = nullptr;
  PackedParams pp;
  std::string err;
  int device = 0;
  hipStream_t stream = nullptr;
  // forward pass / reverse sweep (their own pairs: the fused call reads both at its end)
  hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr;
  bool defer_forward_sync = false;                 // dfx_forward_kinetic_value_and_grad: the forward pass returns without waiting for the device
  dfx_stats fwd_stats;                             // ... its statistics, completed by finish_forward
  bool have_params = false, have_traj = false, have_fields = false;
  bool use_graph = true;
  bool want_bond_grads = true, want_fn_grads = true, want_damping_grads = true;
  DevBuf<int32_t> d_slot_info, d_block_special, d_slot_bond, d_touch, d_ovf_ptr, d_ovf_info, d_ovf_bond;
  DevBuf<double> d_ovf_p, d_ovf_g;             // extra ligaments (general bond lists): parameters, gradient accumulators
  DevBuf<double> d_out_r, d_out_phi, d_out_lam;    // gradients re-laid-out on the device (collect_grads)
  DevBuf<double> d_resp;                           // dfx_response_data outputs
  bool device_views = false;                       // this call hands out device pointers (dfx_kinetic_value_and_grad_device)
  DevBuf<dfx_special> d_special;
  DevBuf<double> d_p_r, d_p_l, d_p_k, d_p_phi, d_cst, d_inv_m, d_damping, d_l_dict, d_p_c, d_g_c;
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
  DevBuf<double> d_POS, d_VEL, d_A, d_state0, d_fields, d_fn_tab, d_restart;
  // segments level, the re-run of piece k-1 beside the reverse stages of piece k (run_adjoint): the second record buffer, hand-off ring and
  // time-function table, and the events that order the two streams (per buffer: records rebuilt / reverse stages done)
  DevBuf<double> d_traj2, d_ring2, d_fn_tab2;
  hipEvent_t ev_rebuilt[2] = {nullptr, nullptr}, ev_reversed[2] = {nullptr, nullptr};
  DevBuf<double> d_YB, d_LAM, d_W, d_KQ, d_G, d_g_r, d_g_phi, d_g_b, d_blk_m, d_blk_c, d_fn_g, d_tmp, d_obj;
  DevBuf<int32_t> d_target;
  std::vector<double> ts;
  std::vector<int> spis;           // RK steps in each output interval (fixed grid)
  std::vector<long long> step0;    // first step ordinal of each interval
  DevBuf<int> d_step_counts;
  int n_counts = 0;
  DevBuf<double> d_acc_times, d_tsteps;
  bool dense = false;              // the last fixed-grid forward kept the stage checkpoint (stage accelerations of every step)
  bool segments = false;           // ... or nothing but the outputs: the reverse sweep re-runs one output interval at a time (records level inside it)
  long long seg_chunk = 0;         // segments level: stage records of at most this many steps are resident (0: a whole output interval); longer intervals are
                                   // re-run in pieces from restart states that the forward pass leaves in d_restart (pieces, below)
  std::vector<int> seg_first, seg_last;   // first / last segment of every output interval
  // segments level with seg_chunk > 0: the pieces of every interval (runs of segments whose stage records are resident together) and, for every
  // piece that does not start its interval, the row of d_restart ([member][row][q | v]) the forward pass leaves its start state in
  struct Piece { int first, last, interval, row; };
  std::vector<Piece> pieces;
  int n_restart_rows = 0;
  bool records = false;            // ... or the records checkpoint (every stage record of every step): no rebuild, no recompute
  std::vector<double> t_steps;     // caller-chosen step boundaries (empty: equal steps); one grid, or one per member (ts_stride = n_total + 1)
  long long ts_stride = 0;
  std::vector<long long> accepted_per_member;
  bool have_adaptive_record = false;
  // dfx_forward_adaptive_keep: the last forward pass was an adaptive solve that kept its accepted steps (stage records in the trajectory
  // checkpoint at the records level, every member's own count) -- the reverse sweep is run_adjoint_dense (dfx_dense.h)
  bool adaptive_records = false;
  long long a_cap = 0, a_stride = 0, a_nmax = 0;      // steps the buffers hold per member, row stride of t_steps / out_ptr, largest N_m
  DevBuf<int> d_out_ptr, d_nacc;
  DevBuf<double> d_theta, d_dw;
  long long n_total = 0;
  std::map<std::pair<int, int>, hipGraphExec_t> graphs;
  // what the cached graphs have baked in: every kernel argument (the DevCtx passed by value) and the addresses the tick node
  // reads and writes (segment table, cursors) -- any of them changing (a buffer re-allocated by a larger solve) drops the graphs
  struct GraphKey { DevCtx ctx; const void* segs; const void* seg_idx; const void* cur; int pair_fwd, pair_adj, pair_rows, pad; } graph_key;
  bool graph_ctx_valid = false;
  // the adaptive controller's graph of 32 attempts (small lattices only), valid for the arguments it was captured with
  hipGraphExec_t adaptive_exec = nullptr;
  struct AdaptiveKey { DevCtx ctx; int n_timepoints; int n_partials; double two_n_free; int keep, pad; AdaptRec ar; } adaptive_key;
  long long launches = 0;
  // two stages per launch on lattice windows (dfx_pair.h): the row length found at create, or tiling_ok = false
#ifdef DFX_EXPERIMENTAL
  TileCtx tile;
#endif
  bool tiling_ok = false;
  int pair_rows = 16;            // window rows = wavefronts per workgroup (16: 1024 threads, 8: 512)
  bool pair_fwd = false, pair_adj = false;   // what the current solve launches (decided per solve: pair_plan)
  // every ligament evaluated once on lattice tiles (dfx_tile.h): the lane tables found at create, the ligament-major images of the
  // parameters (k_lig_pack after every set_params) and of the node-vector / void-angle accumulators (k_lig_unpack after a sweep)
  bool wt = false;               // the table builds of the stage kernels store write-through (sc1): launches that fill the chip (dfx_create)
  bool stage_builds = true;      // ... and take their per-stage builds (DFX_STAGE_BUILDS=0: the generic ones, for A/B runs)
  bool lig_ok = false, lig_used = false;      // lig_used: accumulators of the running sweep are ligament-major
  bool lig_fwd_used = false, lig_adj_used = false;   // what the last forward pass / reverse sweep launched (dfx_stats)
#ifdef DFX_EXPERIMENTAL
  LigCtx lig;
#endif
  // the stage loop without kernel boundaries (dfx_persist.h): decided per solve (persist_plan); the hand-off ring
  bool persist_fwd = false, persist_adj = false;
  int persist_npb = 4, persist_wpm = 0, n_cu = 0;
  int persist_fwd_members = 0, persist_adj_members = 0;     // members per launch (the rest follow in further launches of the same segment)
  DevBuf<double> d_ring, d_err3;       // d_err3: the per-wave partials of the adaptive controller's error norm (dfx_persist_dense.h)
  int spin_limit = 0;                  // > 0: polls before a wave of a persistent launch gives up (dfx_test_set_spin_limit; 0: kSpinLimit)
  // failure isolation (problems/quads_focusing_multi_input.py:66-77: a member that diverges yields NaN for itself only):
  // status of every member after the last forward pass -- 0 ok, 1 non-finite state, 2 step size underflow, 3 step budget exceeded --
  // and whether such a member fails the call (default) or is merely flagged (dfx_set_failure_policy)
  std::vector<int32_t> member_status;
  bool isolate_failures = false;
  bool persist_off = false;            // a wave of a persistent launch gave up once (a workgroup was not resident): this handle keeps one launch per
                                       // stage from then on (the solve that met it was re-run that way, in the same process)
  std::vector<int32_t> lig_slots;
  DevBuf<int32_t> d_lig_slots, d_lig_tab;
  DevBuf<double> d_lig_p, d_lig_l, d_lig_k, d_lig_phi, d_lig_g, d_lig_gphi;
};

enum { kCkState = 0, kCkStages = 1, kCkRecords = 2, kCkSegments = 3 };      // what the forward pass keeps for the reverse sweep (engine_forward.hip)
static const char* const kPersistGaveUp =
    "a wave of the persistent stage loop gave up waiting for a neighbour's record (a workgroup of the launch was not resident: "
    "another process on the device?); DFX_PERSIST=0 keeps one launch per stage";

// internal functions that cross translation units: hidden, they are not part of the library's interface (include/dfx.h is)
#pragma GCC visibility push(hidden)
// engine_launch.hip
int pair_state_buf(const dfx_handle* h, long long n);
void drop_graphs(dfx_handle* h);
DevCtx make_ctx(dfx_handle* h);
dim3 slot_grid(const dfx_handle* h);
dim3 slot_grid(const dfx_handle* h, const Group& g);
DevCtx group_ctx(const dfx_handle* h, const DevCtx& c, int gi);
StageCoef stage_coef(const Tableau& T, int i);
AdjCoef adj_coef(const Tableau& T, int i);
void launch_fn_table(dfx_handle* h, const DevCtx& c, hipStream_t st, int nm, int n_steps);
void setup_tiling(dfx_handle* h);
void pair_plan(dfx_handle* h, const DevCtx& c);
void setup_lig(dfx_handle* h);
int lig_pack(dfx_handle* h);
bool lig_fwd_ok(const dfx_handle* h, const DevCtx& c, int mode);
bool lig_adj_ok(const dfx_handle* h, const DevCtx& c, int wbuf, int local_only);
bool pack3(const dfx_handle* h);
int kernel_build_code(const dfx_handle* h, const DevCtx& c, bool tile);
void launch_fwd(dfx_handle* h, const DevCtx& c, hipStream_t st, dim3 grid, int i, int j, int in_buf, int out_buf, int y_buf, int mode);
void launch_fwd(dfx_handle* h, const DevCtx& c, int i, int j, int in_buf, int out_buf, int y_buf, int mode);
void launch_adj(dfx_handle* h, const DevCtx& c, hipStream_t st, dim3 grid, int i, int j, int in_buf, int wbuf, int local_only);
void launch_adj(dfx_handle* h, const DevCtx& c, int i, int j, int in_buf, int wbuf, int local_only);
bool persist_shape_ok(const dfx_handle* h);
bool persist_would_serve(dfx_handle* h);
int persist_members_that_fit(dfx_handle* h, const void* fn, int npb);
bool persist_members_ok(const dfx_handle* h, int per_launch);
void persist_plan(dfx_handle* h, const DevCtx& c);
void persist_plan_adj(dfx_handle* h, const DevCtx& c);
int* persist_give_up_word(dfx_handle* h);
int ensure_flags(dfx_handle* h);               // the pinned flag words: [0] non-finite (any member), [1] touched, [2] give-up, [16 + m] non-finite (member m)
int* member_flags(dfx_handle* h);
int persist_spin_limit(const dfx_handle* h);
void persist_forget(dfx_handle* h);            // dfx_destroy: drop the handle's streams from the account of persistent launches in flight
void persist_fell_back(dfx_handle* h);         // a launch could not get its workgroups resident: latch the handle onto stage launches, say so once
bool persist_adaptive_plan(dfx_handle* h);
void launch_adaptive_persist(dfx_handle* h, const DevCtx& c, hipStream_t st, int max_attempts, AdaptLoopArgs aa);
bool persist_plan_adj_dense(dfx_handle* h, const DevCtx& c);
void launch_adj_dense_persist(dfx_handle* h, const DevCtx& c, hipStream_t st, int n_steps, DenseCtx dn);
int step_units(const dfx_handle* h, int kind);
bool use_fn_table(const dfx_handle* h);
bool solve_is_eager(const dfx_handle* h);
void enqueue_interleaved(dfx_handle* h, const DevCtx& cbase, int n_steps, int kind, int seg_index = -1);
bool seg_overlap_plan(dfx_handle* h, const DevCtx& c);
void enqueue_rerun_segment(dfx_handle* h, const DevCtx& cf, hipStream_t st, int seg_index);
int run_segment(dfx_handle* h, const DevCtx& c, int gi, int n_steps, int kind);
int fork_groups(dfx_handle* h);
int join_groups(dfx_handle* h);
// engine_forward.hip
int choose_checkpoint(dfx_handle* h, long long n_steps, long long max_interval_steps);
int ensure_work_buffers(dfx_handle* h);
int finish_forward(dfx_handle* h, dfx_stats* stats);
int forward_grid_impl(dfx_handle* h, const double* state0, const double* timepoints, int32_t n_timepoints, const int32_t* steps_per_interval,
    const double* step_times, int32_t keep_trajectory, double* fields, dfx_stats* stats, bool per_member);
// engine_dense.hip: the reverse sweep of an adaptive solve that kept its accepted steps
int run_adjoint_dense(dfx_handle* h, const dfx_grads* want, dfx_grads* grads, dfx_grads* views, dfx_stats* stats, bool kinetic, int n_target);
// engine_reverse.hip
int ensure_adjoint_buffers(dfx_handle* h);
int zero_grad_accumulators(dfx_handle* h, double* extra = nullptr, size_t n_extra = 0, int cursor_value = -1, const int32_t* targets = nullptr,
    int n_target = 0);
int collect_grads(dfx_handle* h, const dfx_grads* want, dfx_grads* grads, dfx_grads* views, bool with_state0);
void set_grad_wishes(dfx_handle* h, const dfx_grads* g);
#pragma GCC visibility pop
