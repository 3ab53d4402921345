// dfx_engine.hip -- libdfx: the MI355X (gfx950) engine behind include/dfx.h (host side; kernels in dfx_kernels.h).
//
// Execution model
//   * one lane per (block, node slot): 4 lanes = one rigid unit, 16 units per 64-wide wavefront, 128-thread workgroups,
//     grid = (ceil(4*n_blocks/128), members of the group).  Every ligament is evaluated by both of its end lanes
//     ("gather form"); the 4 slot contributions of a unit are summed with DPP quad moves; lanes 0..2 of the quad own DOF
//     x, y, theta in the integrator epilogue.  No LDS; atomics only for the (rare) time-function parameter gradients that
//     several DOFs of one block share: results are reproducible run to run.
//   * one kernel launch per Runge-Kutta stage (the neighbour exchange of an explicit stage is a grid-wide dependency; a
//     kernel boundary is the cheapest grid barrier on this chip).  The epilogue of stage i assembles the
//     stage record of stage i+1 (48 B per unit: x y th cos(th/2) sin(th/2) pad, plus 24 B of velocity); a lane gathers its
//     partner's record from a guessed slot in the same batch of loads as everything else.
//   * the time loop is replayed from hipGraphs (one graph = one segment of <= kMaxGraphSteps steps, per member group and
//     direction); what changes between replays (time, step size, checkpoint slot) is one 64-byte record in device memory,
//     refreshed by a 1-thread tick kernel at the head of each graph.  Member groups advance on concurrent streams.
//   * the reverse sweep reads the checkpointed trajectory (72 B per unit and step) and runs one dual-number kernel per
//     stage; the stage records it linearises about are either rebuilt elementwise from the stage checkpoint (stage
//     accelerations of every step, kept when they fit in HBM) or recomputed by forward launches.
//
// The per-ligament physics (dfx_physics.h) is shared with the CPU port of the oracle.
#define DFX_ABI_LAYOUT_IMPL      // include/dfx.h then carries the body of dfx_abi_layout
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
#include "dfx_pair.h"
#include "dfx_tile.h"

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
static size_t dfx_test_free_bytes = 0;     // DFX_TEST_FREE_BYTES: pretend that only so much HBM is free (tests of the checkpoint choice)

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
  PinnedBuf stage, obj_stage, flag_stage, zero_phi;   // flag_stage: one-word results (non-finite flag, touched flag) land in pinned memory; zero_phi: an all-zero void-angle gradient handed out when the sweep never touched the accumulator
  hipEvent_t ev_fork = nullptr;
  PackedParams pp;
  std::string err;
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr;   // forward pass / reverse sweep (their own pairs: the fused call reads both at its end)
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
  DevBuf<double> d_POS, d_VEL, d_A, d_state0, d_fields, d_fn_tab;
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
  std::vector<int> seg_first, seg_last;   // first / last segment of every output interval
  bool records = false;            // ... or the records checkpoint (every stage record of every step): no rebuild, no recompute
  std::vector<double> t_steps;     // caller-chosen step boundaries (empty: equal steps); one grid, or one per member (ts_stride = n_total + 1)
  long long ts_stride = 0;
  std::vector<long long> accepted_per_member;
  bool have_adaptive_record = false;
  long long n_total = 0;
  std::map<std::pair<int, int>, hipGraphExec_t> graphs;
  // what the cached graphs have baked in: every kernel argument (the DevCtx passed by value) and the addresses the tick node
  // reads and writes (segment table, cursors) -- any of them changing (a buffer re-allocated by a larger solve) drops the graphs
  struct GraphKey { DevCtx ctx; const void* segs; const void* seg_idx; const void* cur; int pair_fwd, pair_adj, pair_rows, pad; } graph_key;
  bool graph_ctx_valid = false;
  // the adaptive controller's graph of 32 attempts (small lattices only), valid for the arguments it was captured with
  hipGraphExec_t adaptive_exec = nullptr;
  struct AdaptiveKey { DevCtx ctx; int n_timepoints; int n_partials; double two_n_free; } adaptive_key;
  long long launches = 0;
  // two stages per launch on lattice windows (dfx_pair.h): the row length found at create, or tiling_ok = false
  TileCtx tile;
  bool tiling_ok = false;
  int pair_rows = 16;            // window rows = wavefronts per workgroup (16: 1024 threads, 8: 512)
  bool pair_fwd = false, pair_adj = false;   // what the current solve launches (decided per solve: pair_plan)
  // every ligament evaluated once on lattice tiles (dfx_tile.h): the lane tables found at create, the ligament-major images of the
  // parameters (k_lig_pack after every set_params) and of the node-vector / void-angle accumulators (k_lig_unpack after a sweep)
  bool wt = false;               // the table builds of the stage kernels store write-through (sc1): launches that fill the chip (dfx_create)
  bool stage_builds = true;      // ... and take their per-stage builds (DFX_STAGE_BUILDS=0: the generic ones, for A/B runs)
  bool lig_ok = false, lig_used = false;      // lig_used: accumulators of the running sweep are ligament-major
  bool lig_fwd_used = false, lig_adj_used = false;   // what the last forward pass / reverse sweep launched (dfx_stats)
  LigCtx lig;
  // the stage loop without kernel boundaries (dfx_persist.h): decided per solve (persist_plan); the hand-off ring
  bool persist_fwd = false, persist_adj = false;
  int persist_npb = 4, persist_wpm = 0, n_cu = 0;
  int persist_fwd_members = 0, persist_adj_members = 0;     // members per launch (the rest follow in further launches of the same segment)
  DevBuf<double> d_ring;
  std::vector<int32_t> lig_slots;
  DevBuf<int32_t> d_lig_slots, d_lig_tab;
  DevBuf<double> d_lig_p, d_lig_l, d_lig_k, d_lig_phi, d_lig_g, d_lig_gphi;
};

static void drop_graphs(dfx_handle* h) {
  for (auto& kv : h->graphs) (void)hipGraphExecDestroy(kv.second);
  h->graphs.clear();
  h->graph_ctx_valid = false;
  if (h->adaptive_exec) { (void)hipGraphExecDestroy(h->adaptive_exec); h->adaptive_exec = nullptr; }
}

static DevCtx make_ctx(dfx_handle* h) {
  const Plan& pl = h->pl;
  DevCtx c;
  memset(&c, 0, sizeof(c));
  c.n_blocks = pl.n_blocks; c.n_slots = pl.n_slots; c.n_fns = pl.n_fns; c.batch = pl.batch; c.s = pl.tab.s;
#ifdef DFX_ABLATE      // experiment builds only: no environment variable changes what the production library computes
  { const char* a = getenv("DFX_ABLATE"); c.ablate = a ? atoi(a) : 0; }
#endif
  c.n_wg = (pl.n_slots + kThreads - 1) / kThreads;
  c.n_wg3 = (pl.n_blocks + (kThreads / 16) * 5 - 1) / ((kThreads / 16) * 5);
  for (int k = 0; k < 4; ++k) c.pred[k] = pl.pred_delta[k];
  c.nbuf = 2 * pl.tab.s;
  c.n_special = pl.n_special; c.k_uniform = h->pp.k_uniform ? 1 : 0; c.n_timepoints = (int)h->ts.size();
  c.slot_info = h->d_slot_info.p; c.block_special = h->d_block_special.p; c.special = h->d_special.p;
  c.n_ovf = pl.n_ovf;
  if (pl.n_ovf) { c.ovf_ptr = h->d_ovf_ptr.p; c.ovf_info = h->d_ovf_info.p; c.ovf_p = h->d_ovf_p.p; c.ovf_g = h->d_ovf_g.p; }
  c.p_lidx = h->d_l_idx.p; c.l_dict = h->d_l_dict.p; c.l_dict_on = h->pp.l_dict_ok ? 1 : 0; c.damping_uniform = h->pp.damping_uniform ? 1 : 0;
  { const char* e = getenv("DFX_DICT_LDS"); c.l_dict_lds = (h->pp.l_dict_ok && h->pp.n_dict_max <= kDictLds && !(e && e[0] == '0')) ? 1 : 0; }
  c.p_r = h->d_p_r.p; c.p_l = h->d_p_l.p; c.p_k = h->d_p_k.p; c.p_phi = h->d_p_phi.p; c.cst = h->d_cst.p;
  c.inv_m = h->d_inv_m.p; c.damping = h->d_damping.p; c.fns = h->d_fns.p;
  c.p_c = h->d_p_c.p; c.g_c = h->d_g_c.p; c.n_npb = pl.n_npb;
  c.cur = h->d_cur.p;
  c.fn_tab = nullptr;          // fixed-grid solves switch it on (use_fn_table): the table is refreshed per segment
  c.clock = h->adaptive ? h->d_clock.p : nullptr;
  c.err_partial = h->d_err_partial.p; c.ts_dev = h->d_ts.p; c.fields_dev = h->d_fields.p;
  c.step_counts = h->adaptive ? h->d_step_counts.p : nullptr;
  c.acc_times = h->adaptive ? h->d_acc_times.p : nullptr; c.acc_cap = kAccCap;
  c.t_steps = (!h->adaptive && !h->t_steps.empty()) ? h->d_tsteps.p : nullptr;
  c.ts_stride = c.t_steps ? h->ts_stride : 0;
  c.rtol = h->rtol; c.atol = h->atol;
  c.traj = h->have_traj ? h->ck->traj.p : nullptr;
  c.rps = (h->have_traj && (h->records || h->segments)) ? pl.tab.s : 1;
  c.AD = (h->have_traj && h->dense && !h->records && !h->segments) ? h->ck->AD.p : nullptr;
  c.ad_stride = pl.batch ? (long long)(h->ck->AD.n / pl.batch) : 0;
  c.POS = h->d_POS.p; c.VEL = h->d_VEL.p; c.A = h->d_A.p;
  c.YB = h->d_YB.p; c.LAM = h->d_LAM.p; c.W = h->d_W.p; c.KQ = h->d_KQ.p; c.G = h->d_G.p;
  c.g_r = h->d_g_r.p; c.g_phi = h->d_g_phi.p; c.g_b = (h->want_bond_grads || pl.n_ovf) ? h->d_g_b.p : nullptr;     // (general bond lists run the one reverse build that has them)
  c.touch = h->d_touch.p;
  c.lam_pairs = (c.g_b || c.AD) ? 0 : 1;      // the REBUILD builds of the reverse stage (launch_adj_t) keep the scalar layout
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
  for (int j = i + 1; j < T.s; ++j) ac.cur[j] = T.a[j][i];
  ac.cur[T.s] = T.a[T.s][i];
  ac.c_i = T.c[i];
  return ac;
}

// ---- lattice windows for the pair launches --------------------------------------------------------------------------------------
// A row length R with block = row * R + col such that every ligament joins blocks at most one row and one column apart.  Candidates
// come from the block offsets the bond list contains; any R that passes is valid (R only shapes the windows, correctness does not
// depend on which one is taken).  Lattices of a single row, or connectivity that is not a grid in the caller's block order, keep the
// one-stage launches.
static bool find_tiling(const Plan& pl, int& R_out) {
  std::map<int, int> deltas;
  for (int s = 0; s < pl.n_slots; ++s)
    if (pl.slot_info[s] >= 0) ++deltas[std::abs((pl.slot_info[s] >> 3) - (s >> 2))];
  std::vector<int> cand;
  for (auto& kv : deltas) if (kv.first > 1) { cand.push_back(kv.first - 1); cand.push_back(kv.first); cand.push_back(kv.first + 1); }
  std::sort(cand.begin(), cand.end());
  for (int R : cand) {
    if (R < 2 || pl.n_blocks % R || pl.n_blocks / R < 2) continue;
    bool ok = true;
    for (int s = 0; s < pl.n_slots && ok; ++s) {
      if (pl.slot_info[s] < 0) continue;
      const int b = s >> 2, pb = pl.slot_info[s] >> 3;
      ok = std::abs(b / R - pb / R) <= 1 && std::abs(b % R - pb % R) <= 1;
    }
    if (ok) { R_out = R; return true; }
  }
  return false;
}
static int tiles_along(int n, int w) { return n <= w ? 1 : 1 + (n - w + (w - 3)) / (w - 2); }
static void setup_tiling(dfx_handle* h) {
  memset(&h->tile, 0, sizeof(h->tile));
  int R = 0;
  h->tiling_ok = find_tiling(h->pl, R);
  if (!h->tiling_ok) return;
  h->tile.R = R; h->tile.n_rows = h->pl.n_blocks / R;
  if (const char* e = getenv("DFX_PAIR_ROWS")) h->pair_rows = atoi(e) == 8 ? 8 : 16;
  else h->pair_rows = h->tile.n_rows <= 8 ? 8 : 16;
  h->tile.tiles_x = tiles_along(R, kWCols);
  h->tile.tiles_y = tiles_along(h->tile.n_rows, h->pair_rows);
  h->tile.n_tiles = h->tile.tiles_x * h->tile.tiles_y;
}
// which launches the next solve uses.  The pair kernels cover: even stage count, no distance-based contact, fixed grid, one ligament
// per node; reverse: the records checkpoint without per-ligament gradients.  Measured (profiles/r03_pair_launches.txt): a pair launch
// is one 1024-thread workgroup per compute unit whose 16 waves load, evaluate, meet at the barrier and evaluate again in lock step, so
// memory time and arithmetic no longer overlap between workgroups, and the window's outer ring adds 15 - 26 % of arithmetic to kernels
// whose vector ALUs are already busy half of the time: 16 x 128x128 forward pair 53 us against 2 x 19.4 us, reverse pair 98 - 138 us
// against 2 x 32.8 us.  For ONE 128x128 system (launch-bound) the forward pairs were ahead (10.5 against 12.1 ms per 250 steps) until
// the stage kernels stopped evaluating time functions in their tails (k_fn_table): 28.6 us per step against 29.1 us.  So the pair
// launches are opt-in: DFX_PAIR=1 both directions where they apply, f / a one direction; the GPU tests keep them exercised.
static void pair_plan(dfx_handle* h, const DevCtx& c) {
  const char* e = getenv("DFX_PAIR");
  const bool can = h->tiling_ok && (h->pl.tab.s % 2 == 0) && h->pl.contact != DFX_CONTACT_DISTANCE && !h->adaptive;
  h->pair_fwd = can && e && (e[0] == '1' || e[0] == 'f');
  h->pair_adj = can && e && (e[0] == '1' || e[0] == 'a') && c.rps > 1 && !c.g_b;
}
static TileCtx group_tile(const dfx_handle* h, int nm) { TileCtx t = h->tile; t.total_wg = t.n_tiles * nm; return t; }

// ---- every ligament once, on lattice tiles (dfx_tile.h) -------------------------------------------------------------------------------
// Ownership: the end on the lower block id.  Lane e = 0 of a block takes the ligament to block b + 1 (same lattice row), lane e = 1 the
// one to the row above at column offset dc1 (one value per lattice: quads 0, kagome -1).  A lattice whose bond list does not fit that
// pattern (a block owning two ligaments in one direction, diagonals both ways, extra ligaments per node) keeps the slot kernels.
static void setup_lig(dfx_handle* h) {
  h->lig_ok = false;
  memset(&h->lig, 0, sizeof(h->lig));
  const Plan& pl = h->pl;
  // Measured (profiles/r04_tile_kernels.txt): correct, 0.65 x the vector instructions of the slot kernels, the same bytes -- and SLOWER
  // (16 x 128x128: forward 21.4 against 18.3 us, reverse 40.2 against 32.9 us): a tile workgroup is a longer chain (loads, barrier,
  // ligaments, barrier, epilogue) on fewer, larger units of work, the reverse kernel needs 208 registers (2 waves per SIMD).  Opt-in
  // (DFX_TILE=1), kept exercised by the GPU tests.
  { const char* e = getenv("DFX_TILE"); if (!(e && e[0] == '1')) return; }
  if (!h->tiling_ok || pl.n_ovf || pl.contact == DFX_CONTACT_DISTANCE) return;
  const int R = h->tile.R, nb = pl.n_blocks;
  std::vector<int32_t> ls((size_t)4 * nb, -1);
  int dc1 = 99;
  for (int s = 0; s < pl.n_slots; ++s) {
    const int info = pl.slot_info[s];
    if (info < 0) continue;
    const int ps = info >> 1, b = s >> 2, pb = ps >> 2;
    if (pb < b) continue;
    int e;
    if (pb == b + 1 && (b % R) + 1 < R) e = 0;
    else {
      const int dr = pb / R - b / R, dc = pb % R - b % R;
      if (dr != 1 || dc < -1 || dc > 1) return;
      if (dc1 == 99) dc1 = dc; else if (dc1 != dc) return;
      e = 1;
    }
    if (ls[(size_t)(2 * b + e) * 2] != -1) return;
    ls[(size_t)(2 * b + e) * 2] = s; ls[(size_t)(2 * b + e) * 2 + 1] = ps;
  }
  h->lig.R = R; h->lig.n_rows = nb / R; h->lig.dc1 = dc1 == 99 ? 0 : dc1;
  h->lig.tiles_x = (R + kTW - 1) / kTW;
  h->lig.n_tiles = h->lig.tiles_x * ((h->lig.n_rows + kTH - 1) / kTH);
  h->lig.n_wg = (h->lig.n_tiles + kTileWaves - 1) / kTileWaves;
  h->lig.inv_tiles_x = (unsigned)((0x100000000ull + (unsigned long long)h->lig.tiles_x - 1) / (unsigned long long)h->lig.tiles_x);   // exact for tile < 2^32 / tiles_x
  if (h->lig.tiles_x < 2) return;            // (2^32 / 1 does not fit the multiplier; a lattice one tile wide gains nothing anyway)
  h->lig_slots.swap(ls);
  h->lig_ok = true;
}
// the images the tile kernels read, from the slot-major ones set_params has just uploaded (same stream)
static int lig_pack(dfx_handle* h) {
  if (!h->lig_ok) return 0;
  const Plan& pl = h->pl;
  const size_t B = pl.batch, n2 = (size_t)pl.n_blocks * 2;
  if (!h->d_lig_slots.p) {
    HIP_OK(h->d_lig_slots.ensure(n2 * 2));
    HIP_OK(hipMemcpyAsync(h->d_lig_slots.p, h->lig_slots.data(), sizeof(int32_t) * n2 * 2, hipMemcpyHostToDevice, h->stream));
  }
  DevCtx c = make_ctx(h);
  const bool need_l = !c.l_dict_lds, need_k = !c.k_uniform, need_phi = pl.contact == DFX_CONTACT_ANGLE;
  HIP_OK(h->d_lig_tab.ensure(B * n2));
  HIP_OK(h->d_lig_p.ensure(B * n2 * 4));
  if (need_l) HIP_OK(h->d_lig_l.ensure(B * n2 * 2));
  if (need_k) HIP_OK(h->d_lig_k.ensure(B * n2 * 4));
  if (need_phi) HIP_OK(h->d_lig_phi.ensure(B * n2 * 2));
  HIP_OK(h->d_lig_g.ensure(B * n2 * 4));
  HIP_OK(h->d_lig_gphi.ensure(B * n2 * 2));
  hipLaunchKernelGGL(k_lig_pack, dim3((unsigned)((n2 + kThreads - 1) / kThreads), (unsigned)B), dim3(kThreads), 0, h->stream, c,
                     (const int32_t*)h->d_lig_slots.p, h->d_lig_tab.p, h->d_lig_p.p, need_l ? h->d_lig_l.p : (double*)nullptr,
                     need_k ? h->d_lig_k.p : (double*)nullptr, need_phi ? h->d_lig_phi.p : (double*)nullptr);
  h->lig.tab = h->d_lig_tab.p; h->lig.p = h->d_lig_p.p; h->lig.l = need_l ? h->d_lig_l.p : nullptr; h->lig.k = need_k ? h->d_lig_k.p : nullptr;
  h->lig.phi = need_phi ? h->d_lig_phi.p : nullptr; h->lig.g = h->d_lig_g.p; h->lig.gphi = h->d_lig_gphi.p;
  return 0;
}
// which launches may take the tile kernels: fixed grid with the segment's time-function table (or no time function at all)
static bool lig_fwd_ok(const dfx_handle* h, const DevCtx& c, int mode) {
  return h->lig_ok && h->lig.tab && !c.clock && !(mode & 2) && (c.fn_tab || h->pl.n_fns == 0);
}
static bool lig_adj_ok(const dfx_handle* h, const DevCtx& c, int wbuf, int local_only) {
  return h->lig_ok && h->lig.tab && !c.clock && (c.fn_tab || h->pl.n_fns == 0) && !c.g_b && !c.AD && !local_only && wbuf < 0;
}

template <int MODEL, int CONTACT>
static void launch_fwd_pair_t(dfx_handle* h, const DevCtx& c, hipStream_t st, int nm, int i, int j, int in_buf, int mid_buf, int out_buf, int y_buf, int mode) {
  const TileCtx tc = group_tile(h, nm);
  const StageCoef s0 = stage_coef(h->pl.tab, i), s1 = stage_coef(h->pl.tab, i + 1);
  if (h->pair_rows == 16) hipLaunchKernelGGL((k_fwd_pair<MODEL, CONTACT, 16>), dim3(tc.total_wg), dim3(1024), 0, st, c, tc, s0, s1, i, j, in_buf, mid_buf, out_buf, y_buf, mode);
  else hipLaunchKernelGGL((k_fwd_pair<MODEL, CONTACT, 8>), dim3(tc.total_wg), dim3(512), 0, st, c, tc, s0, s1, i, j, in_buf, mid_buf, out_buf, y_buf, mode);
}
static void launch_fwd_pair(dfx_handle* h, const DevCtx& c, hipStream_t st, int nm, int i, int j, int in_buf, int mid_buf, int out_buf, int y_buf, int mode) {
  const Plan& pl = h->pl;
#define DFX_FP_CASE(M) case M: if (pl.contact) launch_fwd_pair_t<M, 1>(h, c, st, nm, i, j, in_buf, mid_buf, out_buf, y_buf, mode); else launch_fwd_pair_t<M, 0>(h, c, st, nm, i, j, in_buf, mid_buf, out_buf, y_buf, mode); break;
  switch (pl.model) { DFX_FP_CASE(kNonlinear) DFX_FP_CASE(kLinearized) DFX_FP_CASE(kSimpleSpring) DFX_FP_CASE(kStretchTorsion) }
#undef DFX_FP_CASE
  h->launches++;
}
template <int MODEL, int CONTACT>
static void launch_adj_pair_t(dfx_handle* h, const DevCtx& c, hipStream_t st, int nm, int i, int j) {
  const TileCtx tc = group_tile(h, nm);
  const AdjCoef a1 = adj_coef(h->pl.tab, i), a2 = adj_coef(h->pl.tab, i - 1);
  if (h->pair_rows == 16) hipLaunchKernelGGL((k_adj_pair<MODEL, CONTACT, 16>), dim3(tc.total_wg), dim3(1024), 0, st, c, tc, a1, a2, i, j);
  else hipLaunchKernelGGL((k_adj_pair<MODEL, CONTACT, 8>), dim3(tc.total_wg), dim3(512), 0, st, c, tc, a1, a2, i, j);
}
static void launch_adj_pair(dfx_handle* h, const DevCtx& c, hipStream_t st, int nm, int i, int j) {
  const Plan& pl = h->pl;
#define DFX_AP_CASE(M) case M: if (pl.contact) launch_adj_pair_t<M, 1>(h, c, st, nm, i, j); else launch_adj_pair_t<M, 0>(h, c, st, nm, i, j); break;
  switch (pl.model) { DFX_AP_CASE(kNonlinear) DFX_AP_CASE(kLinearized) DFX_AP_CASE(kSimpleSpring) DFX_AP_CASE(kStretchTorsion) }
#undef DFX_AP_CASE
  h->launches++;
}

// 3-node blocks: the two stage kernels pack five triangles per 16 lanes (lane_pos<3>) instead of leaving every fourth lane idle --
// 20 blocks per wave instead of 16.  Fixed grid only (the adaptive controller's error reduction keeps the quad mapping), not with the
// distance-based contact; reverse: the records build.  DFX_PACK3=0 keeps the quad mapping (A/B measurements).
static bool pack3(const dfx_handle* h) {
  const char* e = getenv("DFX_PACK3");
  return h->pl.n_npb == 3 && !h->adaptive && !(e && e[0] == '0');
}
// The per-stage builds of the stage kernels (template parameter ISTAGE) assume the common parameter shape and compile its run-time flags
// away: uniform stiffnesses and damping, the reference-vector dictionary in LDS, equal steps, no stage checkpoint, no adaptive clock,
// records read from / written to the trajectory checkpoint (the caller checks the buffer arguments).
static bool hot_shape(const DevCtx& c) {
  return c.k_uniform && c.l_dict_on && c.l_dict_lds && c.damping_uniform && !c.t_steps && !c.AD && !c.clock;
}
// the per-stage builds (stage index, common parameter shape and -- where the records live in the checkpoint -- the buffer arguments as
// compile-time constants) of the write-through table kernels, quad mapping (NPB = 4) or packed triangles (NPB = 3); false: not applicable
template <int MODEL, int CONTACT, int NPB>
static bool launch_fwd_hot(dfx_handle* h, const DevCtx& c, hipStream_t st, dim3 grid, int i, int j, int in_buf, int out_buf, int y_buf, int mode) {
  if constexpr ((MODEL == kNonlinear || MODEL == kLinearized) && CONTACT != 2) {
    if (!h->wt || !h->stage_builds || !hot_shape(c)) return false;
    const StageCoef scf = stage_coef(h->pl.tab, i);
    const bool recs = c.rps > 1 && in_buf == -1 - i && out_buf == -2 - i && y_buf == -1 && mode == 0;
#define DFX_FWD_I(I) case I: if (recs) hipLaunchKernelGGL((k_fwd_stage<MODEL, CONTACT, NPB, 1, 0, 1, I, 1>), grid, dim3(kThreads), 0, st, c, scf, i, j, in_buf, out_buf, y_buf, mode); \
                             else hipLaunchKernelGGL((k_fwd_stage<MODEL, CONTACT, NPB, 1, 0, 1, I, 0>), grid, dim3(kThreads), 0, st, c, scf, i, j, in_buf, out_buf, y_buf, mode); return true;
    switch (i) { DFX_FWD_I(0) DFX_FWD_I(1) DFX_FWD_I(2) DFX_FWD_I(3) DFX_FWD_I(4) DFX_FWD_I(5) default: break; }
#undef DFX_FWD_I
  }
  return false;
}
template <int MODEL, int CONTACT, int NPB>
static bool launch_adj_hot(dfx_handle* h, const DevCtx& c, hipStream_t st, dim3 grid, int i, int j, int in_buf, int wbuf, int local_only, const StageCoef& rc, int rb) {
  if constexpr ((MODEL == kNonlinear || MODEL == kLinearized) && CONTACT != 2) {
    if (!h->wt || !h->stage_builds || !(c.k_uniform && c.l_dict_on && c.l_dict_lds && c.damping_uniform && !c.t_steps)) return false;
    const AdjCoef acf = adj_coef(h->pl.tab, i);
#define DFX_ADJ_I(I) case I: hipLaunchKernelGGL((k_adj_stage<MODEL, CONTACT, 0, 0, NPB, 1, 0, 1, I>), grid, dim3(kThreads), 0, st, c, acf, i, j, in_buf, wbuf, local_only, rc, rb); return true;
    switch (i) { DFX_ADJ_I(0) DFX_ADJ_I(1) DFX_ADJ_I(2) DFX_ADJ_I(3) DFX_ADJ_I(4) DFX_ADJ_I(5) default: break; }
#undef DFX_ADJ_I
  }
  return false;
}
// what dfx_stats.tile_kernels reports: 1 tile kernels, 2 the per-stage / common-shape builds of the slot kernels, 0 their generic builds
static int kernel_build_code(const dfx_handle* h, const DevCtx& c, bool tile) {
  if (tile) return 1;
  const bool model_ok = (h->pl.model == kNonlinear || h->pl.model == kLinearized) && h->pl.contact != DFX_CONTACT_DISTANCE;
  const bool quad_or_packed = !h->pl.n_ovf && (h->pl.n_npb == 4 || pack3(h));
  return (model_ok && quad_or_packed && h->wt && h->stage_builds && c.fn_tab && hot_shape(c) && h->pl.tab.s <= 6) ? 2 : 0;
}
template <int MODEL, int CONTACT>
static void launch_fwd_t(dfx_handle* h, const DevCtx& c, hipStream_t st, dim3 grid, int i, int j, int in_buf, int out_buf, int y_buf, int mode) {
  if constexpr (CONTACT != 2) {
    if (lig_fwd_ok(h, c, mode)) {
      const dim3 tg(h->lig.n_wg, grid.y);
      hipLaunchKernelGGL((k_fwd_tile<MODEL, CONTACT>), tg, dim3(64 * kTileWaves), 0, st, c, h->lig, stage_coef(h->pl.tab, i), i, j, in_buf, out_buf, y_buf, mode);
      return;
    }
  }
  if (h->pl.n_ovf) {       // general bond lists: the build that walks a node's extra ligaments (quad mapping, in-kernel time functions)
    if constexpr (CONTACT != 2)
      hipLaunchKernelGGL((k_fwd_stage<MODEL, CONTACT, 4, 0, 1>), grid, dim3(kThreads), 0, st, c, stage_coef(h->pl.tab, i), i, j, in_buf, out_buf, y_buf, mode);
    return;
  }
  const bool tab = c.fn_tab != nullptr && !c.clock;        // the segment's time-function table is there: the build that reads it
  if (CONTACT != 2 && pack3(h) && !(mode & 2)) {
    if constexpr (CONTACT != 2) {
      if (tab && launch_fwd_hot<MODEL, CONTACT, 3>(h, c, st, dim3(c.n_wg3, grid.y), i, j, in_buf, out_buf, y_buf, mode)) return;
      if (tab) hipLaunchKernelGGL((k_fwd_stage<MODEL, CONTACT, 3, 1>), dim3(c.n_wg3, grid.y), dim3(kThreads), 0, st, c, stage_coef(h->pl.tab, i), i, j, in_buf, out_buf, y_buf, mode);
      else hipLaunchKernelGGL((k_fwd_stage<MODEL, CONTACT, 3, 0>), dim3(c.n_wg3, grid.y), dim3(kThreads), 0, st, c, stage_coef(h->pl.tab, i), i, j, in_buf, out_buf, y_buf, mode);
    }
    return;
  }
  if (tab && h->wt) {
    if (launch_fwd_hot<MODEL, CONTACT, 4>(h, c, st, grid, i, j, in_buf, out_buf, y_buf, mode)) return;
    hipLaunchKernelGGL((k_fwd_stage<MODEL, CONTACT, 4, 1, 0, 1>), grid, dim3(kThreads), 0, st, c, stage_coef(h->pl.tab, i), i, j, in_buf, out_buf, y_buf, mode);
  }
  else if (tab) hipLaunchKernelGGL((k_fwd_stage<MODEL, CONTACT, 4, 1>), grid, dim3(kThreads), 0, st, c, stage_coef(h->pl.tab, i), i, j, in_buf, out_buf, y_buf, mode);
  else hipLaunchKernelGGL((k_fwd_stage<MODEL, CONTACT, 4, 0>), grid, dim3(kThreads), 0, st, c, stage_coef(h->pl.tab, i), i, j, in_buf, out_buf, y_buf, mode);
}
static void launch_fwd(dfx_handle* h, const DevCtx& c, hipStream_t st, dim3 grid, int i, int j, int in_buf, int out_buf, int y_buf, int mode) {
  const Plan& pl = h->pl;
#define DFX_FWD_CASE(M) case M: if (pl.contact == 2) launch_fwd_t<M, 2>(h, c, st, grid, i, j, in_buf, out_buf, y_buf, mode); else if (pl.contact) launch_fwd_t<M, 1>(h, c, st, grid, i, j, in_buf, out_buf, y_buf, mode); else launch_fwd_t<M, 0>(h, c, st, grid, i, j, in_buf, out_buf, y_buf, mode); break;
  switch (pl.model) { DFX_FWD_CASE(kNonlinear) DFX_FWD_CASE(kLinearized) DFX_FWD_CASE(kSimpleSpring) DFX_FWD_CASE(kStretchTorsion) }
#undef DFX_FWD_CASE
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
  if constexpr (CONTACT != 2) {
    if (lig_adj_ok(h, c, wbuf, local_only)) {
      const dim3 tg(h->lig.n_wg, grid.y);
      hipLaunchKernelGGL((k_adj_tile<MODEL, CONTACT>), tg, dim3(64 * kTileWaves), 0, st, c, h->lig, adj_coef(h->pl.tab, i), i, j, in_buf);
      return;
    }
  }
  if (h->pl.n_ovf) {       // general bond lists: one build for every checkpoint level (per-ligament gradients on, rebuild on)
    if constexpr (CONTACT != 2)
      hipLaunchKernelGGL((k_adj_stage<MODEL, CONTACT, 1, 1, 4, 0, 1>), grid, dim3(kThreads), 0, st, c, adj_coef(h->pl.tab, i), i, j, in_buf, wbuf, local_only, rc, rb);
  }
  else if (c.g_b) hipLaunchKernelGGL((k_adj_stage<MODEL, CONTACT, 1, 1>), grid, dim3(kThreads), 0, st, c, adj_coef(h->pl.tab, i), i, j, in_buf, wbuf, local_only, rc, rb);
  else if (c.AD) hipLaunchKernelGGL((k_adj_stage_rb<MODEL, CONTACT>), grid, dim3(kThreads), 0, st, c, adj_coef(h->pl.tab, i), i, j, in_buf, wbuf, local_only, rc, rb);
  else if (CONTACT != 2 && pack3(h)) {
    if constexpr (CONTACT != 2) {
      if (c.fn_tab && !local_only && launch_adj_hot<MODEL, CONTACT, 3>(h, c, st, dim3(c.n_wg3, grid.y), i, j, in_buf, wbuf, local_only, rc, rb)) return;
      if (c.fn_tab && !local_only) hipLaunchKernelGGL((k_adj_stage<MODEL, CONTACT, 0, 0, 3, 1>), dim3(c.n_wg3, grid.y), dim3(kThreads), 0, st, c, adj_coef(h->pl.tab, i), i, j, in_buf, wbuf, local_only, rc, rb);
      else hipLaunchKernelGGL((k_adj_stage<MODEL, CONTACT, 0, 0, 3, 0>), dim3(c.n_wg3, grid.y), dim3(kThreads), 0, st, c, adj_coef(h->pl.tab, i), i, j, in_buf, wbuf, local_only, rc, rb);
    }
  }
  else if (c.fn_tab && !local_only && h->wt) {
    if (launch_adj_hot<MODEL, CONTACT, 4>(h, c, st, grid, i, j, in_buf, wbuf, local_only, rc, rb)) return;
    hipLaunchKernelGGL((k_adj_stage<MODEL, CONTACT, 0, 0, 4, 1, 0, 1>), grid, dim3(kThreads), 0, st, c, adj_coef(h->pl.tab, i), i, j, in_buf, wbuf, local_only, rc, rb);
  }
  else if (c.fn_tab && !local_only) hipLaunchKernelGGL((k_adj_stage<MODEL, CONTACT, 0, 0, 4, 1>), grid, dim3(kThreads), 0, st, c, adj_coef(h->pl.tab, i), i, j, in_buf, wbuf, local_only, rc, rb);
  else hipLaunchKernelGGL((k_adj_stage<MODEL, CONTACT, 0, 0, 4, 0>), grid, dim3(kThreads), 0, st, c, adj_coef(h->pl.tab, i), i, j, in_buf, wbuf, local_only, rc, rb);
}
static void launch_adj(dfx_handle* h, const DevCtx& c, hipStream_t st, dim3 grid, int i, int j, int in_buf, int wbuf, int local_only) {
  const Plan& pl = h->pl;
#define DFX_ADJ_CASE(M) case M: if (pl.contact == 2) launch_adj_t<M, 2>(h, c, st, grid, i, j, in_buf, wbuf, local_only); else if (pl.contact) launch_adj_t<M, 1>(h, c, st, grid, i, j, in_buf, wbuf, local_only); else launch_adj_t<M, 0>(h, c, st, grid, i, j, in_buf, wbuf, local_only); break;
  switch (pl.model) { DFX_ADJ_CASE(kNonlinear) DFX_ADJ_CASE(kLinearized) DFX_ADJ_CASE(kSimpleSpring) DFX_ADJ_CASE(kStretchTorsion) }
#undef DFX_ADJ_CASE
  h->launches++;
}
static void launch_adj(dfx_handle* h, const DevCtx& c, int i, int j, int in_buf, int wbuf, int local_only) {
  launch_adj(h, c, h->stream, slot_grid(h), i, j, in_buf, wbuf, local_only);
}

// ---- the stage loop without kernel boundaries (dfx_persist.h) ---------------------------------------------------------------------
// Two persistent launches must never share the device: each needs ALL its workgroups resident, and two half-resident launches would
// wait for each other until their spins give up.  Every persistent launch of the process therefore waits for the one before it
// (whatever handle or stream issued it) through one event per device.
using namespace dfx_persist;
static const char* kPersistGaveUp = "a wave of the persistent stage loop gave up waiting for a neighbour's record (a workgroup of the launch was not resident: "
                                    "another process on the device?); DFX_PERSIST=0 keeps one launch per stage";
static std::mutex g_persist_mu;
static hipEvent_t g_persist_tail[64];
static bool g_persist_tail_on[64];
static const int kPersistLdsBudget = 150 * 1024;    // of a compute unit's 160 KB: room for the stage kernels' small LDS users next to us

// workgroups of `fn` a compute unit can hold at once (registers; 256-thread workgroups = one wave per SIMD each), capped where the
// residency rule of MI355X_MICROARCH.md ("Residency and cooperative launch") starts to depend on the scalar-register count
static int persist_wg_per_cu(const void* fn) {
  hipFuncAttributes at;
  if (!fn || hipFuncGetAttributes(&at, fn) != hipSuccess) { (void)hipGetLastError(); return 0; }
  const int alloc = std::max(8, ((at.numRegs + 7) / 8) * 8);
  int cap = std::min({8, 512 / alloc, 6});
  if (const char* e = getenv("DFX_PERSIST_MAX_WG")) cap = std::min(cap, atoi(e));
  return cap;
}
// which lattices and solves the persistent kernels serve (everything else keeps one launch per stage)
static bool persist_shape_ok(const dfx_handle* h) {
  const Plan& pl = h->pl;
  const char* e = getenv("DFX_PERSIST");
  if (e && e[0] == '0') return false;
  return (pl.model == kNonlinear || pl.model == kLinearized) && pl.contact != DFX_CONTACT_DISTANCE && !pl.n_ovf && pl.tab.s <= kPersistStages &&
         (pl.n_npb == 3 || pl.n_npb == 4);
}
static int persist_waves_per_member(const dfx_handle* h, int npb) {
  return npb == 3 ? (h->pl.n_blocks + 19) / 20 : (h->pl.n_slots + 63) / 64;
}
// how many members fit on the chip at once (0: not even one), and the launch shape for `nm` of them
static int persist_members_that_fit(dfx_handle* h, const void* fn, int npb) {
  if (!h->n_cu) { int v = 0; if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, h->device) != hipSuccess || v <= 0) return 0; h->n_cu = v; }
  const int cap = persist_wg_per_cu(fn);
  const long long wpm = persist_waves_per_member(h, npb);
  if (cap <= 0 || wpm <= 0) return 0;
  return (int)std::min<long long>(h->pl.batch, ((long long)cap * h->n_cu * 4) / wpm);
}
static void persist_shape(const dfx_handle* h, int npb, int nm, int* grid, int* lds) {
  const long long waves = (long long)nm * persist_waves_per_member(h, npb);
  const long long g = (waves + 3) / 4;
  const long long per_cu = std::max<long long>(1, (g + h->n_cu - 1) / h->n_cu);
  *grid = (int)g;
  *lds = (kPersistLdsBudget / (int)per_cu) & ~1023;
}
// launches per segment a solve may be cut into (members that do not fit at once follow in further launches of the same segment); beyond
// it the stage launches serve the solve: a launch that fills the chip several times over is what the stage kernels are tuned for.
// Measured (profiles/r05_persistent_kernels.txt): 8 designs of the 64x64-cell kagome lattice -- forward in one launch 5.3 against 6.9 us
// per stage, reverse in two launches of 4 designs 8.5 against 10.7 us; 16 x 128x128 cut into 4 + 8 launches: 22 / 34 against 14 / 26 us.
static int persist_max_chunks() {
  const char* e = getenv("DFX_PERSIST_CHUNKS");
  return e ? std::max(1, atoi(e)) : 2;
}
static bool persist_common_ok(dfx_handle* h, const DevCtx& c) {
  if (!persist_shape_ok(h) || h->adaptive || h->groups.size() != 1) return false;
  if (h->pl.n_fns > 0 && !c.fn_tab) return false;
  h->persist_npb = (h->pl.n_npb == 3 && pack3(h)) ? 3 : 4;
  h->persist_wpm = persist_waves_per_member(h, h->persist_npb);
  return true;
}
static bool persist_members_ok(const dfx_handle* h, int per_launch) {
  return per_launch > 0 && (h->pl.batch + per_launch - 1) / per_launch <= persist_max_chunks();
}
// decided per solve, after the context is known
static void persist_plan(dfx_handle* h, const DevCtx& c) {
  h->persist_fwd = false;
  if (h->pair_fwd || h->lig_fwd_used || !persist_common_ok(h, c)) return;
  const void* fn = dfx_persist::fwd_kernel(h->pl.model, h->pl.contact, h->persist_npb);
  if (!fn) return;
  h->persist_fwd_members = persist_members_that_fit(h, fn, h->persist_npb);
  if (!persist_members_ok(h, h->persist_fwd_members)) return;
  if (h->d_ring.ensure((size_t)kPRing * h->pl.batch * h->pl.n_blocks * kPos) != hipSuccess) { (void)hipGetLastError(); return; }
  h->persist_fwd = true;
}
static void persist_plan_adj(dfx_handle* h, const DevCtx& c) {
  h->persist_adj = false;
  if (h->pair_adj || h->lig_adj_used || !persist_common_ok(h, c)) return;
  if (c.rps <= 1 || c.g_b || c.AD || !c.lam_pairs) return;          // the records build of the reverse stage, nothing else
  const void* fn = dfx_persist::adj_kernel(h->pl.model, h->pl.contact, h->persist_npb);
  if (!fn) return;
  h->persist_adj_members = persist_members_that_fit(h, fn, h->persist_npb);
  if (!persist_members_ok(h, h->persist_adj_members)) return;
  if (h->d_ring.ensure((size_t)kPRing * h->pl.batch * h->pl.n_blocks * kPos) != hipSuccess) { (void)hipGetLastError(); return; }
  h->persist_adj = true;
}
static PersistCoef persist_coef(const Tableau& T) {
  PersistCoef pc;
  memset(&pc, 0, sizeof(pc));
  for (int i = 0; i < T.s && i < kPersistStages; ++i)
    for (int l = 0; l <= i; ++l) { pc.cv[i][l] = T.a[i + 1][l]; pc.cq[i][l] = T.aa[i + 1][l]; }
  for (int r = 0; r <= T.s && r <= kPersistStages; ++r) pc.c[r] = T.c[r];
  return pc;
}
static int* persist_give_up_word(dfx_handle* h) { return reinterpret_cast<int*>(h->flag_stage.p) + 2; }   // (word 0: non-finite flag, word 1: touched flag)
// one launch, chained behind the previous persistent launch of the process
static void launch_persist(dfx_handle* h, const void* fn, hipStream_t st, void** args, int grid, int lds) {
  static std::map<const void*, int> lds_set;
  std::lock_guard<std::mutex> lk(g_persist_mu);
  if (lds_set[fn] < lds) { (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, kPersistLdsBudget); lds_set[fn] = kPersistLdsBudget; }
  const int d = h->device & 63;
  if (g_persist_tail_on[d]) (void)hipStreamWaitEvent(st, g_persist_tail[d], 0);
  else { (void)hipEventCreateWithFlags(&g_persist_tail[d], hipEventDisableTiming); g_persist_tail_on[d] = true; }
  (void)hipLaunchKernel(fn, dim3(grid), dim3(kPersistThreads), args, lds, st);
  (void)hipEventRecord(g_persist_tail[d], st);
  h->launches++;
}
// one segment of the group's members: the first ring places poisoned, then the whole segment in one launch per `per_launch` members
static void launch_segment_persist(dfx_handle* h, const DevCtx& c, hipStream_t st, int nm, int n_steps, bool reverse) {
  const int npb = h->persist_npb, per = reverse ? h->persist_adj_members : h->persist_fwd_members;
  const void* fn = reverse ? dfx_persist::adj_kernel(h->pl.model, h->pl.contact, npb) : dfx_persist::fwd_kernel(h->pl.model, h->pl.contact, npb);
  PersistCoef pcf = persist_coef(h->pl.tab);
  PersistAdjCoef pca;
  memset(&pca, 0, sizeof(pca));
  for (int i = 0; i < h->pl.tab.s && i < kPersistStages; ++i) {
    const AdjCoef ac = adj_coef(h->pl.tab, i);
    for (int jj = 0; jj <= kPersistStages; ++jj) { pca.col[i][jj] = ac.col[jj]; pca.cur[i][jj] = ac.cur[jj]; }
    pca.c[i] = ac.c_i;
  }
  for (int off = 0; off < nm; off += per) {
    const int cnt = std::min(per, nm - off);
    DevCtx cc = c;
    cc.m0 = c.m0 + off;
    int grid = 0, lds = 0;
    persist_shape(h, npb, cnt, &grid, &lds);
    dfx_persist::launch_ring_poison(st, h->d_ring.p, h->pl.batch, h->pl.n_blocks, cc.m0, cnt, kPos);
    h->launches++;
    PersistArgs pa;
    pa.ring = h->d_ring.p; pa.give_up = persist_give_up_word(h); pa.n_steps = n_steps; pa.nm = cnt; pa.waves_per_member = h->persist_wpm; pa.pad = 0;
    void* args_f[] = {&cc, &pcf, &pa};
    void* args_r[] = {&cc, &pca, &pa};
    launch_persist(h, fn, st, reverse ? args_r : args_f, grid, lds);
  }
}
static void launch_fwd_persist(dfx_handle* h, const DevCtx& c, hipStream_t st, int nm, int n_steps) { launch_segment_persist(h, c, st, nm, n_steps, false); }
static void launch_adj_persist(dfx_handle* h, const DevCtx& c, hipStream_t st, int nm, int n_steps) { launch_segment_persist(h, c, st, nm, n_steps, true); }

// forward: stage i reads buffer fin(i), writes fout(i); buffer 0 is the step state
static int fin(int i) { return i == 0 ? 0 : 1 + ((i - 1) & 1); }
static int fout(int i, int s) { return i == s - 1 ? 0 : 1 + (i & 1); }
// one forward stage of step j of the segment: with the records checkpoint the records live in the trajectory only (stage i reads
// record i of step n and writes record i+1; record s of step n is the state of step n+1), else in the ping-pong stage buffers
static void launch_fwd_step_stage(dfx_handle* h, const DevCtx& c, hipStream_t st, dim3 grid, int i, int j) {
  const int s = h->pl.tab.s;
  if (c.rps > 1) launch_fwd(h, c, st, grid, i, j, -1 - i, -1 - (i + 1), -1, 0);
  else launch_fwd(h, c, st, grid, i, j, fin(i), fout(i, s), 0, (i == s - 1 && c.traj) ? 1 : 0);
}
static int adj_in_buf(const DevCtx& c, int i) { return c.rps > 1 ? -1 - i : (i == 0 ? -1 : i); }
// A step is s one-stage launches or s / 2 pair launches ("units"); unit u of step j, forward / reverse:
static int step_units(const dfx_handle* h, int kind) { return (kind == 0 ? h->pair_fwd : h->pair_adj) ? h->pl.tab.s / 2 : h->pl.tab.s; }
static void launch_fwd_unit(dfx_handle* h, const DevCtx& c, hipStream_t st, dim3 grid, int u, int j) {
  if (!h->pair_fwd) { launch_fwd_step_stage(h, c, st, grid, u, j); return; }
  const int s = h->pl.tab.s, i = 2 * u, nm = (int)grid.y;
  // records checkpoint: every record lives in the trajectory; otherwise the records ping-pong between stage buffers 1 and 2 and the
  // step state between buffers 0 and 3 (k_fwd_pair resolves buffer 0 by the parity of the step)
  if (c.rps > 1) launch_fwd_pair(h, c, st, nm, i, j, -1 - i, -1 - (i + 1), -1 - (i + 2), -1, 0);
  else launch_fwd_pair(h, c, st, nm, i, j, u == 0 ? 0 : 1 + ((u - 1) & 1), -1, i + 2 == s ? 0 : 1 + (u & 1), 0, (i + 2 == s && c.traj) ? 1 : 0);
}
// reverse unit u counts from the END of the step (u = 0: the last stage / pair)
static void launch_adj_unit(dfx_handle* h, const DevCtx& c, hipStream_t st, dim3 grid, int u, int j) {
  const int s = h->pl.tab.s;
  if (!h->pair_adj) { const int i = s - 1 - u; launch_adj(h, c, st, grid, i, j, adj_in_buf(c, i), -1, 0); return; }
  launch_adj_pair(h, c, st, (int)grid.y, s - 1 - 2 * u, j);
}

// the time functions of the segment the group's cursor now points at, for every step and stage time (k_fn_table); DFX_FN_TABLE=0:
// the lanes of driven / loaded blocks evaluate them themselves, as in rounds 1-2
static bool use_fn_table(const dfx_handle* h) {
  const char* e = getenv("DFX_FN_TABLE");
  return h->pl.n_fns > 0 && !h->adaptive && h->d_fn_tab.p && !(e && e[0] == '0');
}
static void launch_fn_table(dfx_handle* h, const DevCtx& c, hipStream_t st, int nm, int n_steps) {
  if (!c.fn_tab) return;
  StageTimes tms;
  for (int r = 0; r < kFnRows; ++r) tms.c[r] = r <= h->pl.tab.s ? h->pl.tab.c[r] : 0.0;
  const int total = n_steps * (h->pl.tab.s + 1) * h->pl.n_fns;
  hipLaunchKernelGGL(k_fn_table, dim3((total + 63) / 64, nm), dim3(64), 0, st, c, tms, n_steps, h->d_fn_tab.p);
  h->launches++;
}

// enqueue one segment (kind 0: forward steps; kind 1: reverse steps) of group gi on that group's stream
static void enqueue_segment(dfx_handle* h, const DevCtx& cbase, int gi, int n_steps, int kind) {
  const int s = h->pl.tab.s;
  const Group& g = h->groups[gi];
  const DevCtx c = group_ctx(h, cbase, gi);
  const dim3 grid = slot_grid(h, g);
  hipLaunchKernelGGL(k_tick, dim3(1), dim3(1), 0, g.stream, (const Seg*)h->d_segs.p, h->d_seg_idx.p + 2 + gi, kind == 0 ? 1 : -1, h->d_cur.p + gi);
  h->launches++;
  launch_fn_table(h, c, g.stream, g.nm, n_steps);
  if (kind == 0 && h->persist_fwd) launch_fwd_persist(h, c, g.stream, g.nm, n_steps);
  else if (kind == 0) {
    for (int j = 0; j < n_steps; ++j)
      for (int u = 0; u < step_units(h, 0); ++u) launch_fwd_unit(h, c, g.stream, grid, u, j);
  } else if (h->persist_adj) launch_adj_persist(h, c, g.stream, g.nm, n_steps);
  else if (c.AD || c.rps > 1) {
    // stage checkpoint: no recompute launches; every reverse launch also rebuilds the record its successor reads
    for (int j = n_steps - 1; j >= 0; --j)
      for (int u = 0; u < step_units(h, 1); ++u) launch_adj_unit(h, c, g.stream, grid, u, j);
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

// Eager launches, the member groups interleaved stage by stage, vs hipGraph replay.  Measured (16 x 128x128, profiles/
// r02_eager_vs_graph.txt): a graph of ~100 nodes takes 0.7 - 2.9 ms from the call to its first kernel -- as long as a 20-step
// solve runs -- and one group's graph launched after the other's staggers the groups by that time; an eager launch costs ~3.5 us of
// host time, less than a stage kernel that fills the chip runs (13 - 45 us), and the first kernel starts at once: eager is
// 10 - 35 % faster up to a few hundred steps and still 1 % faster at 5 000.  Graphs keep the launch cost off the host where the
// kernels are short (small lattices / few members: launch-bound at ~5 us per stage, below the eager launch rate) and the solve is
// long enough to hide the first launch.  Rule: eager when the launches fill the chip (>= 2 waves per SIMD) or the solve is short
// (<= 128 steps); DFX_EAGER_STEPS=<n> overrides the step threshold for every size (0: always graphs).
static bool solve_is_eager(const dfx_handle* h) {
  if (!h->use_graph) return true;
  if (const char* e = getenv("DFX_EAGER_STEPS")) return h->n_total <= atoll(e);
  const long long waves = (long long)h->pl.batch * ((h->pl.n_slots + 63) / 64);
  return waves >= 2048 || h->n_total <= 128;
}

static void enqueue_interleaved(dfx_handle* h, const DevCtx& cbase, int n_steps, int kind, int seg_index = -1) {
  const int ng = (int)h->groups.size();
  if (kind == 1 && !cbase.AD && cbase.rps == 1) {      // recompute chains (events per group): group by group
    for (int gi = 0; gi < ng; ++gi) enqueue_segment(h, cbase, gi, n_steps, kind);
    return;
  }
  std::vector<DevCtx> cg(ng, cbase);
  for (int gi = 0; gi < ng; ++gi) {
    cg[gi] = group_ctx(h, cbase, gi);
    if (seg_index >= 0) hipLaunchKernelGGL(k_set_seg, dim3(1), dim3(1), 0, h->groups[gi].stream, (const Seg*)h->d_segs.p, seg_index, h->d_cur.p + gi);
    else hipLaunchKernelGGL(k_tick, dim3(1), dim3(1), 0, h->groups[gi].stream, (const Seg*)h->d_segs.p, h->d_seg_idx.p + 2 + gi, kind == 0 ? 1 : -1, h->d_cur.p + gi);
    h->launches++;
    launch_fn_table(h, cg[gi], h->groups[gi].stream, h->groups[gi].nm, n_steps);
  }
  if (kind == 0 && h->persist_fwd) {
    for (int gi = 0; gi < ng; ++gi) launch_fwd_persist(h, cg[gi], h->groups[gi].stream, h->groups[gi].nm, n_steps);
  } else if (kind == 0) {
    for (int j = 0; j < n_steps; ++j)
      for (int u = 0; u < step_units(h, 0); ++u)
        for (int gi = 0; gi < ng; ++gi)
          launch_fwd_unit(h, cg[gi], h->groups[gi].stream, slot_grid(h, h->groups[gi]), u, j);
  } else if (h->persist_adj) {
    for (int gi = 0; gi < ng; ++gi) launch_adj_persist(h, cg[gi], h->groups[gi].stream, h->groups[gi].nm, n_steps);
  } else {
    for (int j = n_steps - 1; j >= 0; --j)
      for (int u = 0; u < step_units(h, 1); ++u)
        for (int gi = 0; gi < ng; ++gi)
          launch_adj_unit(h, cg[gi], h->groups[gi].stream, slot_grid(h, h->groups[gi]), u, j);
  }
}

static int run_segment(dfx_handle* h, const DevCtx& c, int gi, int n_steps, int kind) {
  if (!h->use_graph) { enqueue_segment(h, c, gi, n_steps, kind); return 0; }
  dfx_handle::GraphKey gk;
  memset(&gk, 0, sizeof(gk));          // padding bytes take part in the memcmp
  memcpy(&gk.ctx, &c, sizeof(DevCtx));
  gk.ctx.n_timepoints = 0;  // not read by the stage kernels
  gk.segs = h->d_segs.p; gk.seg_idx = h->d_seg_idx.p; gk.cur = h->d_cur.p;
  gk.pair_fwd = h->pair_fwd; gk.pair_adj = h->pair_adj; gk.pair_rows = h->pair_rows;
  if (!h->graph_ctx_valid || memcmp(&h->graph_key, &gk, sizeof(gk)) != 0) {
    drop_graphs(h);
    memcpy(&h->graph_key, &gk, sizeof(gk));
    h->graph_ctx_valid = true;
  }
  auto key = std::make_pair(n_steps, kind * kMaxGroups + gi);
  auto it = h->graphs.find(key);
  const int s = h->pl.tab.s;
  const long long per = 1 + (c.fn_tab ? 1 : 0) + (long long)n_steps * ((kind == 0 || c.AD || c.rps > 1) ? step_units(h, kind) : 2 * s - 1);   // launches in the graph
  hipStream_t st = h->groups[gi].stream;
  if (it == h->graphs.end()) {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    const long long before = h->launches;
    HIP_OK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    enqueue_segment(h, c, gi, n_steps, kind);
    hipError_t ce = hipStreamEndCapture(st, &graph);     // always ends the capture, also after a failed launch inside it
    h->launches = before;
    if (ce == hipSuccess) ce = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    if (graph) (void)hipGraphDestroy(graph);
    if (ce != hipSuccess) { h->err = std::string("graph capture / instantiate: ") + hipGetErrorString(ce); return 2; }
    (void)hipGraphUpload(exec, st);                      // device-side setup now, not inside the first timed replay
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
//
// What the forward pass keeps for the reverse sweep -- three levels, the richest that fits in HBM is taken:
//   records  every stage record of every step (72 s B per unit and step): the reverse launch of stage i reads the record it
//            linearises about straight from the checkpoint -- s launches per step, nothing rebuilt, nothing recomputed; the forward
//            pass writes its records there instead of into the ping-pong buffers, i.e. no extra forward traffic;
//   stages   the step states + the first s-1 stage accelerations of every step (72 + 24 (s-1) B): s launches per step, each
//            reverse launch rebuilds the record its successor reads (elementwise);
//   state    the step states only (72 B): 2s - 1 launches per step (s - 1 forward launches recompute the records);
//   segments nothing but the outputs the solve keeps anyway: the reverse sweep visits the output intervals backwards, re-runs the
//            forward pass of ONE interval from its (resident) output row with the records checkpoint for that interval only, then
//            reverses it: 3s launches per step in all, memory independent of the horizon -- taken when not even the step states fit
//            (the full 50 000-step C3 then runs 16 members per GPU instead of 4).
// DFX_CHECKPOINT=records|stages|state overrides (DFX_STAGE_CHECKPOINT=1/0 = stages / state, kept for older scripts).
enum { kCkState = 0, kCkStages = 1, kCkRecords = 2, kCkSegments = 3 };

static int choose_checkpoint(dfx_handle* h, long long n_steps, long long max_interval_steps) {
  const Plan& pl = h->pl;
  { const char* t = getenv("DFX_TEST_FREE_BYTES"); dfx_test_free_bytes = t ? (size_t)atoll(t) : 0; }
  const size_t B = pl.batch, rec = (size_t)pl.n_blocks * kStep, N = (size_t)std::max<long long>(n_steps, 1);
  const size_t want_rec = B * (N * pl.tab.s + 1) * rec;
  const size_t want_state = B * (N + 1) * rec;
  const size_t want_ad = B * N * (pl.tab.s - 1) * pl.n_blocks * 3;
  int forced = -1;
  if (const char* e = getenv("DFX_CHECKPOINT")) {
    if (!strcmp(e, "records")) forced = kCkRecords;
    else if (!strcmp(e, "stages")) forced = kCkStages;
    else if (!strcmp(e, "state")) forced = kCkState;
    else if (!strcmp(e, "segments")) forced = kCkSegments;
    else {
      static bool warned = false;
      if (!warned) fprintf(stderr, "[dfx] DFX_CHECKPOINT=%s is not one of records|stages|state|segments: ignored\n", e);
      warned = true;
    }
  } else if (const char* e2 = getenv("DFX_STAGE_CHECKPOINT")) forced = e2[0] != '0' ? kCkStages : kCkState;
  const size_t want_seg = B * ((size_t)std::max<long long>(max_interval_steps, 1) * pl.tab.s + 1) * rec;
  // the kernels address trajectory records by a 32-bit ordinal ((step * records per step + record) * members + member; at the segments
  // level the step is the global one against a shifted base): 2^32 records are 44 million Dopri5 steps of 16 members
  if ((double)B * ((double)N * pl.tab.s + 1.0) >= 4294967296.0) {
    h->err = "steps x members too large: the trajectory checkpoint is addressed by 32-bit record ordinals (split the ensemble or the horizon)";
    return -2;
  }
  // a level whose buffers already exist fits whatever else has been allocated since (adjoint work buffers, sibling engines of a
  // multi-input objective, RCCL): only GROWTH is checked against the free memory, leaving 5 % of the device.  The driver is asked
  // for the free memory only when something has to grow (the query costs ~0.1 ms: a repeated solve of the same shape skips it).
  size_t free_b = 0, total_b = 0;
  int have_info = -1;
  auto fits = [&](size_t grow_elems) {
    if (grow_elems == 0) return true;
    if (have_info < 0) have_info = hipMemGetInfo(&free_b, &total_b) == hipSuccess ? 1 : 0;
    const size_t free_now = dfx_test_free_bytes ? std::min<size_t>(free_b, dfx_test_free_bytes) : free_b;
    return have_info == 1 && grow_elems * sizeof(double) + total_b / 20 <= free_now;
  };
  const size_t have_t = h->ck->traj.n, have_a = h->ck->AD.n;
  auto grow = [](size_t want, size_t have) { return want > have ? want - have : (size_t)0; };   // DevBuf frees the old block before it allocates
  int mode = forced;
  if (mode < 0) {
    // records whenever they fit: the reverse launch reads the record it linearises about instead of rebuilding it (128x128 x 16:
    // 32 us against 42 us; 24x16 x 256: forward + reverse 254 + 370 ms against 245 + 473 ms, profiles/r03_c5_shared_checkpoint.txt).
    // Round 2 kept small lattices at the stages level because THREE engines of a multi-input objective each allocated a 50 - 130 GB
    // checkpoint; engines whose inputs run in turn now share one (dfx_share_checkpoint), and a level that does not fit next to what
    // other handles hold falls back by itself.
    if (fits(grow(want_rec, have_t))) mode = kCkRecords;
    else if (fits(grow(want_state, have_t) + grow(want_ad, have_a))) mode = kCkStages;
    else if (fits(grow(want_state, have_t))) mode = kCkState;
    else mode = kCkSegments;
  }
  // A re-allocated (or, after a failed allocation, freed) buffer no longer holds what its last writer put there: handles that share the
  // pool (dfx_share_checkpoint) must not run a reverse sweep on it.  DevBuf::ensure frees before it allocates, so the pointers tell.
  const double* t0 = h->ck->traj.p;
  const double* a0 = h->ck->AD.p;
  auto done = [&](int m) {
    if (h->ck->traj.p != t0 || h->ck->AD.p != a0) h->ck->writer = nullptr;
    return m;
  };
  if (mode == kCkSegments) {
    if (h->ck->traj.ensure(want_seg) != hipSuccess) { (void)hipGetLastError(); return done(-1); }
    return done(mode);
  }
  // allocate; a failed allocation falls back one level (forced modes included: the solve still runs)
  if (mode == kCkRecords && h->ck->traj.ensure(want_rec) != hipSuccess) { (void)hipGetLastError(); mode = kCkStages; }
  if (mode != kCkRecords && h->ck->traj.ensure(want_state) != hipSuccess) { (void)hipGetLastError(); return done(-1); }
  if (mode == kCkStages && h->ck->AD.ensure(want_ad) != hipSuccess) { (void)hipGetLastError(); mode = kCkState; }
  return done(mode);
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
  HIP_OK(h->d_cur.ensure(kMaxGroups));
  if (pl.n_fns > 0) HIP_OK(h->d_fn_tab.ensure(B * (size_t)kMaxGraphSteps * kFnRows * DFX_MAX_FNS * kFnEntry));
  return 0;
}

static int ensure_adjoint_buffers(dfx_handle* h) {
  const Plan& pl = h->pl;
  const size_t B = pl.batch, nb = pl.n_blocks, s = pl.tab.s;
  const size_t nsp = std::max(1, pl.n_special);
  HIP_OK(h->d_YB.ensure(B * s * nb * 6));
  HIP_OK(h->d_LAM.ensure(2 * B * nb * 6));      // x 2: the pair launches double-buffer lambda by step parity
  HIP_OK(h->d_W.ensure(B * 2 * nb * 3));
  HIP_OK(h->d_KQ.ensure(B * 2 * nb * 3));
  HIP_OK(h->d_g_r.ensure(B * pl.n_slots * 2));
  HIP_OK(h->d_g_phi.ensure(B * pl.n_slots));
  HIP_OK(h->d_g_b.ensure(B * pl.n_slots * 8));
  HIP_OK(h->d_blk_m.ensure(B * nb * 3));
  HIP_OK(h->d_blk_c.ensure(B * nb * 3));
  if (pl.contact == DFX_CONTACT_DISTANCE) HIP_OK(h->d_g_c.ensure(B * nb * 2));
  HIP_OK(h->d_fn_g.ensure(B * nsp * DFX_MAX_FNS * DFX_FN_PARAMS));
  return 0;
}

// One launch (k_prelude) clears the gradient accumulators -- and, for the callers that pass them, the cotangent array G, the groups'
// segment cursors (set to `cursor_value`) and the target blocks (copied from the kernel arguments when they are few).
static int zero_grad_accumulators(dfx_handle* h, double* extra = nullptr, size_t n_extra = 0, int cursor_value = -1,
                                  const int32_t* targets = nullptr, int n_target = 0) {
  const Plan& pl = h->pl;
  const size_t B = pl.batch, nb = pl.n_blocks;
  const size_t nsp = std::max(1, pl.n_special);
  PreludeJob J;
  memset(&J, 0, sizeof(J));
  size_t most = 0;
  auto zero = [&](double* p, size_t n) { if (p && n) { J.zp[J.n_zero] = p; J.zn[J.n_zero] = n; ++J.n_zero; most = std::max(most, n); } };
  static_assert(kPreludeZero >= 11, "every accumulator below + one caller array");
  zero(h->d_g_r.p, B * pl.n_slots * 2);
  zero(h->d_g_phi.p, B * pl.n_slots);
  zero(reinterpret_cast<double*>(h->d_touch.p), 2);              // 4 ints
  h->lig_used = false;
  if (h->lig_ok && h->lig.g) { zero(h->d_lig_g.p, B * nb * 8); zero(h->d_lig_gphi.p, B * nb * 4); }
  if (pl.n_ovf) zero(h->d_ovf_g.p, B * pl.n_ovf * kOvfG);
  if (h->want_bond_grads || pl.n_ovf) zero(h->d_g_b.p, B * pl.n_slots * 8);
  zero(h->d_blk_m.p, B * nb * 3);
  if (pl.contact == DFX_CONTACT_DISTANCE) zero(h->d_g_c.p, B * nb * 2);
  if (h->want_damping_grads) zero(h->d_blk_c.p, B * nb * 3);
  zero(h->d_fn_g.p, B * nsp * DFX_MAX_FNS * DFX_FN_PARAMS);
  zero(extra, n_extra);
  if (cursor_value >= 0) { J.fill_dst = h->d_seg_idx.p + 2; J.fill_val = cursor_value; J.fill_n = kMaxGroups; }
  if (targets && n_target > 0) {
    if (n_target <= kPreludeInts) { J.copy_dst = h->d_target.p; J.copy_n = n_target; for (int i = 0; i < n_target; ++i) J.copy_val[i] = targets[i]; }
    else HIP_OK(hipMemcpyAsync(h->d_target.p, targets, sizeof(int32_t) * n_target, hipMemcpyHostToDevice, h->stream));
  }
  const unsigned gx = (unsigned)std::max<size_t>(1, std::min<size_t>(2048, (most / 2 + 255) / 256));
  hipLaunchKernelGGL(k_prelude, dim3(gx, (unsigned)std::max(1, J.n_zero)), dim3(256), 0, h->stream, J);
  return 0;
}

// Gradient accumulators -> the layouts of dfx_grads.  The big ones (node vectors, void angles, inertia, damping, state0) are
// re-laid-out by ONE device kernel, so that what crosses PCIe is final: a single batch of DMA transfers into the pinned staging
// area and no scatter loops on the host (a solve of a few steps is otherwise dominated by this function).  `grads` (caller
// buffers, may be null) receives copies; `views` (may be null) receives pointers INTO the staging area, valid until the next
// call on the handle -- the zero-copy form the Python layer wraps in NumPy arrays.  Entries that are non-null in `want`
// are produced.
static int collect_grads(dfx_handle* h, const dfx_grads* want, dfx_grads* grads, dfx_grads* views, bool with_state0) {
  const Plan& pl = h->pl;
  const size_t B = pl.batch, nb = pl.n_blocks, NS = pl.n_slots, nbd = pl.n_bonds;
  const size_t nsp = std::max(1, pl.n_special);
  const int npb = pl.n_npb;
  if (views) memset(views, 0, sizeof(*views));
  if (!want) {
    HIP_OK(hipStreamSynchronize(h->stream));
    HIP_OK(hipGetLastError());
    return 0;
  }
  if (h->lig_used) {        // the tile kernels accumulated ligament-major: fold into the slot-major accumulators read below
    DevCtx c = make_ctx(h);
    hipLaunchKernelGGL(k_lig_unpack, dim3((unsigned)((nb * 2 + kThreads - 1) / kThreads), (unsigned)B), dim3(kThreads), 0, h->stream, c,
                       (const int32_t*)h->d_lig_slots.p, (const double*)h->d_lig_g.p, pl.contact == DFX_CONTACT_ANGLE ? (const double*)h->d_lig_gphi.p : (const double*)nullptr);
    h->lig_used = false;
  }
  if (h->device_views) {    // the gradients stay where the sweep accumulated them: re-layout on the device, no copy over PCIe
    if (want->reference_vector || want->k_bond || want->contact || want->fn_params || pl.n_ovf ||
        (want->void_angle0 && pl.contact != DFX_CONTACT_ANGLE) || (want->block_centroids && pl.contact != DFX_CONTACT_DISTANCE)) {
      h->err = "device-resident gradients: centroid_node_vectors, void_angle0 (angle contact), inertia, damping, state0, block_centroids "
               "(distance contact) of lattices without extra ligaments; the others are assembled on the host (dfx_kinetic_value_and_grad)";
      return 1;
    }
    const bool d_r = want->centroid_node_vectors, d_phi = want->void_angle0, d_lam = with_state0 && want->state0;
    const bool d_pack_r = d_r && npb != kSlots;
    if (d_pack_r) HIP_OK(h->d_out_r.ensure(B * nb * npb * 2));
    if (d_phi) HIP_OK(h->d_out_phi.ensure(B * nbd * 2));
    if (d_lam) HIP_OK(h->d_out_lam.ensure(B * nb * 6));
    if (d_pack_r || d_phi || d_lam) {
      DevCtx c = make_ctx(h);
      hipLaunchKernelGGL(k_pack_grads, slot_grid(h), dim3(kThreads), 0, h->stream, c, (const int32_t*)h->d_slot_bond.p, npb, (int)nbd,
                         d_pack_r ? h->d_out_r.p : (double*)nullptr, d_phi ? h->d_out_phi.p : (double*)nullptr,
                         d_lam ? h->d_out_lam.p : (double*)nullptr);
    }
    HIP_OK(hipStreamSynchronize(h->stream));
    HIP_OK(hipGetLastError());
    dfx_grads v;
    memset(&v, 0, sizeof(v));
    if (d_r) v.centroid_node_vectors = d_pack_r ? h->d_out_r.p : h->d_g_r.p;
    if (d_phi) v.void_angle0 = h->d_out_phi.p;
    if (want->inertia) v.inertia = h->d_blk_m.p;
    if (want->damping) v.damping = h->d_blk_c.p;
    if (d_lam) v.state0 = h->d_out_lam.p;
    if (want->block_centroids) v.block_centroids = h->d_g_c.p;
    if (views) *views = v;
    return 0;
  }
  const bool w_r = want->centroid_node_vectors;
  bool w_phi = want->void_angle0 && pl.contact == DFX_CONTACT_ANGLE;
  // contacts are rare: when no lane of the sweep added to the void-angle accumulator (one flag, known after the sweep) its gradient
  // is identically zero -- neither re-laid-out nor downloaded (8 MB of the 31 MB that leave the device for 16 x 128x128), the caller
  // gets a view of a zero buffer that is never written
  bool phi_zero = false;
  if (w_phi) {
    HIP_OK(h->flag_stage.ensure(64));
    HIP_OK(hipMemcpyAsync(reinterpret_cast<int32_t*>(h->flag_stage.p) + 1, h->d_touch.p, sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));   // (word 0: the forward pass's flag)
    HIP_OK(hipStreamSynchronize(h->stream));
    const int32_t touched = reinterpret_cast<const int32_t*>(h->flag_stage.p)[1];
    if (!touched && !pl.n_ovf) {
      const size_t bytes = sizeof(double) * B * nbd * 2;
      if (h->zero_phi.n < bytes || !h->zero_phi.p) { HIP_OK(h->zero_phi.ensure(bytes)); memset(h->zero_phi.p, 0, h->zero_phi.n); }
      phi_zero = true; w_phi = false;
    }
  }
  const bool w_cen = want->block_centroids && pl.contact == DFX_CONTACT_DISTANCE;
  const bool w_b = h->want_bond_grads && (want->reference_vector || want->k_bond || want->contact);
  const bool w_m = want->inertia, w_c = want->damping && h->want_damping_grads, w_fn = want->fn_params && h->want_fn_grads;
  const bool w_lam = with_state0 && want->state0;
  const size_t n_r = w_r ? B * nb * npb * 2 : 0, n_phi = w_phi ? B * nbd * 2 : 0, n_b = w_b ? B * NS * 8 : 0, n_m = w_m ? B * nb * 3 : 0,
               n_c = w_c ? B * nb * 3 : 0, n_fn = w_fn ? B * nsp * DFX_MAX_FNS * DFX_FN_PARAMS : 0, n_lam = w_lam ? B * nb * 6 : 0,
               n_cen = w_cen ? B * nb * 2 : 0;
  // small host-side results (bond parameters, time-function parameters) live behind the DMA area
  const size_t n_small = (want->reference_vector ? B * nbd * 2 : 0) + (want->k_bond ? B * nbd * 3 : 0) + (want->contact ? B * 3 : 0) +
                         (want->fn_params ? B * (size_t)std::max(1, pl.n_fns) * DFX_FN_PARAMS : 0) + (want->damping && !w_c ? B * nb * 3 : 0) +
                         (want->void_angle0 && !w_phi && !phi_zero ? B * nbd * 2 : 0) + (want->block_centroids && !w_cen ? B * nb * 2 : 0);
  HIP_OK(h->stage.ensure((n_r + n_phi + n_b + n_m + n_c + n_fn + n_lam + n_cen + n_small + 8) * sizeof(double)));
  double* g_r = reinterpret_cast<double*>(h->stage.p);
  double* g_phi = g_r + n_r;
  double* g_b = g_phi + n_phi;
  double* g_m = g_b + n_b;
  double* g_c = g_m + n_m;
  double* fn_g = g_c + n_c;
  double* lam = fn_g + n_fn;
  double* cen = lam + n_lam;
  double* small = cen + n_cen;
  // device-side re-layout: kagome node vectors (3 of 4 slots), void angles (slot -> (bond, end)), state0 (q | v planes)
  const bool pack_r = w_r && npb != kSlots;
  if (pack_r || w_phi || w_lam) {
    if (pack_r) HIP_OK(h->d_out_r.ensure(n_r));
    if (w_phi) HIP_OK(h->d_out_phi.ensure(n_phi));
    if (w_phi && pl.n_ovf) HIP_OK(hipMemsetAsync(h->d_out_phi.p, 0, sizeof(double) * n_phi, h->stream));   // ends that are extra ligaments: added on the host
    if (w_lam) HIP_OK(h->d_out_lam.ensure(n_lam));
    DevCtx c = make_ctx(h);
    hipLaunchKernelGGL(k_pack_grads, slot_grid(h), dim3(kThreads), 0, h->stream, c, (const int32_t*)h->d_slot_bond.p, npb, (int)nbd,
                       pack_r ? h->d_out_r.p : (double*)nullptr, w_phi ? h->d_out_phi.p : (double*)nullptr,
                       w_lam ? h->d_out_lam.p : (double*)nullptr);
  }
  auto pull = [&](double* dst, const double* src, size_t n) {
    return n ? hipMemcpyAsync(dst, src, sizeof(double) * n, hipMemcpyDeviceToHost, h->stream) : hipSuccess;
  };
  HIP_OK(pull(g_r, pack_r ? h->d_out_r.p : h->d_g_r.p, n_r));
  HIP_OK(pull(g_phi, h->d_out_phi.p, n_phi));
  HIP_OK(pull(g_b, h->d_g_b.p, n_b));
  HIP_OK(pull(g_m, h->d_blk_m.p, n_m));
  HIP_OK(pull(g_c, h->d_blk_c.p, n_c));
  HIP_OK(pull(fn_g, h->d_fn_g.p, n_fn));
  HIP_OK(pull(lam, h->d_out_lam.p, n_lam));
  HIP_OK(pull(cen, h->d_g_c.p, n_cen));
  HIP_OK(hipStreamSynchronize(h->stream));
  HIP_OK(hipGetLastError());
  dfx_grads v;
  memset(&v, 0, sizeof(v));
  if (w_r) v.centroid_node_vectors = g_r;
  if (w_phi) v.void_angle0 = g_phi;
  if (phi_zero) v.void_angle0 = reinterpret_cast<double*>(h->zero_phi.p);
  if (w_m) v.inertia = g_m;
  if (w_c) v.damping = g_c;
  if (w_lam) v.state0 = lam;
  if (w_cen) v.block_centroids = cen;
  auto take = [&](size_t n) { double* q = small; small += n; memset(q, 0, sizeof(double) * n); return q; };
  if (want->void_angle0 && !w_phi && !phi_zero) v.void_angle0 = take(B * nbd * 2);
  if (want->damping && !w_c) v.damping = take(B * nb * 3);
  if (want->block_centroids && !w_cen) v.block_centroids = take(B * nb * 2);
  if (want->reference_vector) v.reference_vector = take(B * nbd * 2);
  if (want->k_bond) v.k_bond = take(B * nbd * 3);
  if (want->contact) v.contact = take(B * 3);
  if (w_b)
    for (size_t m = 0; m < B; ++m)
      for (size_t sl = 0; sl < NS; ++sl) {
        const int info = pl.slot_info[sl];
        if (info < 0 || (info & 1)) continue;            // one entry per ligament: its end-0 slot
        const size_t bond = (size_t)pl.slot_bond[sl];
        const double* q = g_b + (m * NS + sl) * 8;
        if (v.reference_vector) { v.reference_vector[(m * nbd + bond) * 2] = q[0]; v.reference_vector[(m * nbd + bond) * 2 + 1] = q[1]; }
        if (v.k_bond) for (int c = 0; c < 3; ++c) v.k_bond[(m * nbd + bond) * 3 + c] = q[2 + c];
        if (v.contact) for (int c = 0; c < 3; ++c) v.contact[m * 3 + c] += q[5 + c];
      }
  if (pl.n_ovf && (v.void_angle0 || v.reference_vector || v.k_bond || v.contact)) {
    // extra ligaments (general bond lists): a handful of entries, unpacked on the host
    std::vector<double> og((size_t)B * pl.n_ovf * kOvfG);
    HIP_OK(hipMemcpy(og.data(), h->d_ovf_g.p, sizeof(double) * og.size(), hipMemcpyDeviceToHost));
    for (size_t m = 0; m < B; ++m)
      for (int e = 0; e < pl.n_ovf; ++e) {
        const double* q = og.data() + (m * pl.n_ovf + e) * kOvfG;
        const size_t bond = (size_t)pl.ovf_bond[e];
        const int end = pl.ovf_info[e] & 1;
        if (v.void_angle0 && pl.contact == DFX_CONTACT_ANGLE) v.void_angle0[(m * nbd + bond) * 2 + end] += q[0];
        if (end || !h->want_bond_grads) continue;
        if (v.reference_vector) { v.reference_vector[(m * nbd + bond) * 2] = q[1]; v.reference_vector[(m * nbd + bond) * 2 + 1] = q[2]; }
        if (v.k_bond) for (int c = 0; c < 3; ++c) v.k_bond[(m * nbd + bond) * 3 + c] = q[3 + c];
        if (v.contact) for (int c = 0; c < 3; ++c) v.contact[m * 3 + c] += q[6 + c];
      }
  }
  if (want->fn_params) {
    v.fn_params = take(B * (size_t)std::max(1, pl.n_fns) * DFX_FN_PARAMS);
    const int W = DFX_MAX_FNS * DFX_FN_PARAMS;
    for (size_t m = 0; m < B; ++m)
      for (int f = 0; f < pl.n_fns; ++f)
        for (int i = 0; i < DFX_FN_PARAMS; ++i) {
          double acc = 0.0;
          if (w_fn) for (int sidx = 0; sidx < pl.n_special; ++sidx) acc += fn_g[(m * pl.n_special + sidx) * W + f * DFX_FN_PARAMS + i];
          v.fn_params[(m * pl.n_fns + f) * DFX_FN_PARAMS + i] = acc;
        }
  }
  if (grads) {
    auto give = [&](double* dst, const double* src, size_t n) { if (dst && src) memcpy(dst, src, sizeof(double) * n); };
    give(grads->centroid_node_vectors, v.centroid_node_vectors, B * nb * npb * 2);
    give(grads->void_angle0, v.void_angle0, B * nbd * 2);
    give(grads->reference_vector, v.reference_vector, B * nbd * 2);
    give(grads->k_bond, v.k_bond, B * nbd * 3);
    give(grads->contact, v.contact, B * 3);
    give(grads->inertia, v.inertia, B * nb * 3);
    give(grads->damping, v.damping, B * nb * 3);
    give(grads->fn_params, v.fn_params, B * (size_t)std::max(1, pl.n_fns) * DFX_FN_PARAMS);
    give(grads->state0, v.state0, B * nb * 6);
    give(grads->block_centroids, v.block_centroids, B * nb * 2);
  }
  if (views) *views = v;
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

int dfx_abi_layout(int32_t* out, int32_t n) { return dfxabi_fill(out, n); }

const char* dfx_last_error(const dfx_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int dfx_create(const dfx_problem* problem, dfx_handle** out) {
  dfx_handle* h = new dfx_handle();
  auto fail = [&](int rc) { g_create_error = h->err; delete h->ck; delete h; return rc; };
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
  (void)hipEventCreate(&h->ev2);
  (void)hipEventCreate(&h->ev3);
  const char* g = getenv("DFX_NO_GRAPH");
  h->use_graph = !(g && g[0] == '1');
  {
    // member groups on concurrent streams hide the launch boundary of one group behind the work of another, but only
    // when a group still fills the chip: measured best 2 groups at >= 2 waves per SIMD in total (128x128 x 4..16
    // members), 1 group below that (24x16 x 32 members: 4.8 s vs 8.4 s with 4 groups)
    const char* e = getenv("DFX_STREAMS");
    const long long waves = (long long)h->pl.batch * ((h->pl.n_slots + 63) / 64);
    int want = e ? atoi(e) : (problem->streams > 0 ? problem->streams : (waves >= 2048 ? 2 : 1));
    // solves that fit the persistent stage loop (dfx_persist.h) run all their members in ONE launch per segment
    if (!e && problem->streams <= 0 && want > 1 && persist_shape_ok(h)) {
      h->persist_npb = (h->pl.n_npb == 3 && pack3(h)) ? 3 : 4;
      // (both sweeps must fit: a solve whose reverse sweep keeps the stage launches keeps its two member groups too)
      if (persist_members_ok(h, persist_members_that_fit(h, dfx_persist::fwd_kernel(h->pl.model, h->pl.contact, h->persist_npb), h->persist_npb)) &&
          persist_members_ok(h, persist_members_that_fit(h, dfx_persist::adj_kernel(h->pl.model, h->pl.contact, h->persist_npb), h->persist_npb))) want = 1;
    }
    int ng = std::max(1, std::min({want, h->pl.batch, kMaxGroups}));
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
  setup_tiling(h);
  {  // write-through stores in the stage kernels where a launch fills the chip (dfx_kernels.h, stg_m); DFX_WT=0|1 overrides (A/B runs)
    const char* e = getenv("DFX_WT");
    h->wt = e ? (e[0] != '0') : ((long long)h->pl.batch * ((h->pl.n_slots + 63) / 64) >= 2048);
    const char* sb = getenv("DFX_STAGE_BUILDS");
    h->stage_builds = !(sb && sb[0] == '0');
  }
  const Plan& pl = h->pl;
  bool ok = h->d_slot_info.ensure(pl.n_slots) == hipSuccess && h->d_block_special.ensure(pl.n_blocks) == hipSuccess &&
            h->d_slot_bond.ensure(pl.n_slots) == hipSuccess && h->d_touch.ensure(4) == hipSuccess &&
            h->d_special.ensure(std::max(1, pl.n_special)) == hipSuccess && h->d_seg_idx.ensure(2 + kMaxGroups) == hipSuccess &&
            h->d_cur.ensure(kMaxGroups) == hipSuccess;
  if (!ok) { h->err = "hipMalloc (static tables) failed"; return fail(2); }
  (void)hipMemcpy(h->d_slot_info.p, pl.slot_info.data(), sizeof(int32_t) * pl.n_slots, hipMemcpyHostToDevice);
  (void)hipMemcpy(h->d_block_special.p, pl.block_special.data(), sizeof(int32_t) * pl.n_blocks, hipMemcpyHostToDevice);
  (void)hipMemcpy(h->d_slot_bond.p, pl.slot_bond.data(), sizeof(int32_t) * pl.n_slots, hipMemcpyHostToDevice);
  if (pl.n_special)
    (void)hipMemcpy(h->d_special.p, pl.special.data(), sizeof(dfx_special) * pl.n_special, hipMemcpyHostToDevice);
  if (pl.n_ovf) {
    if (h->d_ovf_ptr.ensure(pl.ovf_ptr.size()) != hipSuccess || h->d_ovf_info.ensure(pl.n_ovf) != hipSuccess || h->d_ovf_bond.ensure(pl.n_ovf) != hipSuccess ||
        h->d_ovf_p.ensure((size_t)pl.batch * pl.n_ovf * kOvfParams) != hipSuccess || h->d_ovf_g.ensure((size_t)pl.batch * pl.n_ovf * kOvfG) != hipSuccess) {
      h->err = "hipMalloc (extra-ligament tables) failed"; return fail(2);
    }
    (void)hipMemcpy(h->d_ovf_ptr.p, pl.ovf_ptr.data(), sizeof(int32_t) * pl.ovf_ptr.size(), hipMemcpyHostToDevice);
    (void)hipMemcpy(h->d_ovf_info.p, pl.ovf_info.data(), sizeof(int32_t) * pl.n_ovf, hipMemcpyHostToDevice);
    (void)hipMemcpy(h->d_ovf_bond.p, pl.ovf_bond.data(), sizeof(int32_t) * pl.n_ovf, hipMemcpyHostToDevice);
    h->tiling_ok = false;          // the pair launches keep to one ligament per node
  }
  setup_lig(h);
  *out = h;
  return 0;
}

int dfx_destroy(dfx_handle* h) {
  if (!h) return 0;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  drop_graphs(h);
  h->d_ovf_ptr.release(); h->d_ovf_info.release(); h->d_ovf_bond.release(); h->d_ovf_p.release(); h->d_ovf_g.release();
  h->d_lig_slots.release(); h->d_lig_tab.release(); h->d_lig_p.release(); h->d_lig_l.release(); h->d_lig_k.release(); h->d_lig_phi.release();
  h->d_lig_g.release(); h->d_lig_gphi.release();
  h->d_slot_info.release(); h->d_block_special.release(); h->d_special.release(); h->d_slot_bond.release(); h->d_touch.release(); h->zero_phi.release();
  h->d_out_r.release(); h->d_out_phi.release(); h->d_out_lam.release(); h->d_resp.release();
  h->d_p_r.release(); h->d_p_l.release(); h->d_p_k.release(); h->d_p_phi.release(); h->d_cst.release(); h->d_l_dict.release(); h->d_l_idx.release();
  h->d_inv_m.release(); h->d_damping.release(); h->d_fns.release(); h->d_p_c.release(); h->d_g_c.release();
  for (int f = 0; f < DFX_MAX_FNS; ++f) h->d_fn_table[f].release();
  h->d_segs.release(); h->d_cur.release(); h->d_seg_idx.release(); h->d_clock.release(); h->d_err_partial.release(); h->d_ts.release(); h->d_step_counts.release(); h->d_acc_times.release(); h->d_tsteps.release(); 
  if (--h->ck->users == 0) { h->ck->traj.release(); h->ck->AD.release(); delete h->ck; }
  h->d_ring.release(); h->d_fn_tab.release(); h->d_POS.release(); h->d_VEL.release(); h->d_A.release(); h->d_state0.release(); h->d_fields.release();
  h->d_YB.release(); h->d_LAM.release(); h->d_W.release(); h->d_KQ.release(); h->d_G.release();
  h->d_g_r.release(); h->d_g_phi.release(); h->d_g_b.release(); h->d_blk_m.release(); h->d_blk_c.release(); h->d_fn_g.release();
  h->d_tmp.release(); h->d_obj.release(); h->d_target.release(); h->stage.release(); h->obj_stage.release(); h->flag_stage.release();
  for (auto& gr : h->groups) { for (auto e : gr.ev_a) (void)hipEventDestroy(e); for (auto e : gr.ev_b) (void)hipEventDestroy(e); if (gr.stream2) (void)hipStreamDestroy(gr.stream2); if (gr.done) (void)hipEventDestroy(gr.done); if (gr.stream && gr.stream != h->stream) (void)hipStreamDestroy(gr.stream); }
  if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
  if (h->ev_fork2) (void)hipEventDestroy(h->ev_fork2);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  if (h->ev2) (void)hipEventDestroy(h->ev2);
  if (h->ev3) (void)hipEventDestroy(h->ev3);
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
  if (h->pl.contact == DFX_CONTACT_DISTANCE) {
    HIP_OK(h->d_p_c.ensure(pp.centroid.size()));
    HIP_OK(hipMemcpyAsync(h->d_p_c.p, pp.centroid.data(), sizeof(double) * pp.centroid.size(), hipMemcpyHostToDevice, h->stream));
  }
  if (h->pl.n_ovf) HIP_OK(hipMemcpyAsync(h->d_ovf_p.p, pp.ovf.data(), sizeof(double) * pp.ovf.size(), hipMemcpyHostToDevice, h->stream));
  HIP_OK(h->d_fns.ensure(pp.fns.size()));
  HIP_OK(hipMemcpyAsync(h->d_fns.p, pp.fns.data(), sizeof(TimeFn) * pp.fns.size(), hipMemcpyHostToDevice, h->stream));
  if (lig_pack(h)) return 2;
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
    const int ck_mode = choose_checkpoint(h, max_steps, std::max<long long>(1, max_steps / std::max(1, max_timepoints - 1)));
    if (ck_mode == -2) return 1;
    if (ck_mode < 0) { h->err = "reserve: cannot allocate the trajectory checkpoint"; return 2; }
  }
  (void)rec;
  return 0;
}

int dfx_share_checkpoint(dfx_handle* h, dfx_handle* with) {
  if (!h || !with) return 1;
  if (h->device != with->device) { h->err = "share_checkpoint: the handles live on different devices"; return 1; }
  if (h->ck == with->ck) return 0;
  HIP_OK(hipSetDevice(h->device));
  HIP_OK(hipStreamSynchronize(h->stream));
  drop_graphs(h);                                          // cached graphs hold the old buffer's address
  if (--h->ck->users == 0) { h->ck->traj.release(); h->ck->AD.release(); delete h->ck; }
  h->ck = with->ck;
  h->ck->users++;
  h->have_traj = false;
  return 0;
}

// after the stream has been waited for: the non-finite flag of the forward pass (pinned word 0 of flag_stage) and its statistics
static int finish_forward(dfx_handle* h, dfx_stats* stats) {
  if (*persist_give_up_word(h)) { h->have_traj = false; h->err = std::string("forward: ") + kPersistGaveUp; return 2; }
  const int bad = *reinterpret_cast<const int*>(h->flag_stage.p);
  if (bad) {
    h->have_traj = false;
    h->err = "forward: non-finite state at output " + std::to_string(bad - 1) + " (unstable step size or contact blow-up)";
    return 3;
  }
  if (stats) {
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, h->ev0, h->ev1);
    *stats = h->fwd_stats;
    stats->kernel_ms = ms;
    stats->stage_kernel_us = h->n_total ? 1e3 * ms / (double)(h->n_total * h->pl.tab.s) : 0.0;
  }
  return 0;
}

int dfx_forward(dfx_handle* h, const double* state0, const double* timepoints, int32_t n_timepoints,
                int32_t steps_per_interval, int32_t keep_trajectory, double* fields, dfx_stats* stats) {
  if (n_timepoints < 1 || steps_per_interval < 1) { h->err = "forward: need >= 1 timepoint and >= 1 step per interval"; return 1; }
  std::vector<int32_t> spis((size_t)std::max(0, n_timepoints - 1), steps_per_interval);
  return dfx_forward_grid(h, state0, timepoints, n_timepoints, spis.data(), nullptr, keep_trajectory, fields, stats);
}

static int forward_grid_impl(dfx_handle* h, const double* state0, const double* timepoints, int32_t n_timepoints,
                             const int32_t* steps_per_interval, const double* step_times, int32_t keep_trajectory,
                             double* fields, dfx_stats* stats, bool per_member);

int dfx_forward_grid(dfx_handle* h, const double* state0, const double* timepoints, int32_t n_timepoints,
                     const int32_t* steps_per_interval, const double* step_times, int32_t keep_trajectory,
                     double* fields, dfx_stats* stats) {
  return forward_grid_impl(h, state0, timepoints, n_timepoints, steps_per_interval, step_times, keep_trajectory, fields, stats, false);
}

int dfx_forward_grid_members(dfx_handle* h, const double* state0, const double* timepoints, int32_t n_timepoints,
                             const int32_t* steps_per_interval, const double* step_times, int32_t keep_trajectory,
                             double* fields, dfx_stats* stats) {
  if (!step_times) { h->err = "forward_grid_members: step_times (batch, n_steps + 1) required"; return 1; }
  return forward_grid_impl(h, state0, timepoints, n_timepoints, steps_per_interval, step_times, keep_trajectory, fields, stats, true);
}

// timepoints: (T,) -- or, per_member, (batch, T); step_times: NULL / (n_steps + 1,) -- or, per_member, (batch, n_steps + 1): every
// member integrates on its own time grid (same step COUNTS: the launches are shared).  Row 0's output times fill the segment table
// (used only without step_times).
static int forward_grid_impl(dfx_handle* h, const double* state0, const double* timepoints, int32_t n_timepoints,
                             const int32_t* steps_per_interval, const double* step_times, int32_t keep_trajectory,
                             double* fields, dfx_stats* stats, bool per_member) {
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
  h->ts_stride = per_member ? h->n_total + 1 : 0;
  if (step_times) {
    const size_t n_grids = per_member ? B : 1;
    h->t_steps.assign(step_times, step_times + n_grids * (size_t)(h->n_total + 1));
    for (size_t g = 0; g < n_grids; ++g) {
      const double* tg = h->t_steps.data() + g * (size_t)(h->n_total + 1);
      for (long long n = 0; n < h->n_total; ++n)
        if (!(tg[n + 1] > tg[n])) { h->err = "forward: step_times must be strictly increasing"; return 1; }
      for (int k = 0; k < Tn; ++k)
        if (tg[h->step0[k]] != timepoints[g * (size_t)Tn + k]) { h->err = "forward: step_times must contain every timepoint at the start of its interval"; return 1; }
    }
    HIP_OK(h->d_tsteps.ensure(h->t_steps.size()));
    HIP_OK(hipMemcpyAsync(h->d_tsteps.p, h->t_steps.data(), sizeof(double) * h->t_steps.size(), hipMemcpyHostToDevice, h->stream));
  }
  const auto tw0 = std::chrono::steady_clock::now();
  if (ensure_work_buffers(h)) return 2;
  HIP_OK(h->d_fields.ensure(B * Tn * nb * 6));
  if (getenv("DFX_TIMING"))
    fprintf(stderr, "[dfx] forward: work buffers %.1f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tw0).count());
  h->have_traj = false;
  if (keep_trajectory) {
    long long max_spi = 1;
    for (int v : h->spis) max_spi = std::max<long long>(max_spi, v);
    const auto tc0 = std::chrono::steady_clock::now();
    const int mode = choose_checkpoint(h, h->n_total, max_spi);
    if (getenv("DFX_TIMING"))
      fprintf(stderr, "[dfx] choose_checkpoint: level %d, %.1f ms (traj %.1f GB, AD %.1f GB, shared by %d)\n", mode,
              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tc0).count(),
              h->ck->traj.n * 8e-9, h->ck->AD.n * 8e-9, h->ck->users);
    if (mode == -2) return 1;
    if (mode < 0) {
      h->err = "forward: cannot allocate the trajectory checkpoint (" + std::to_string((B * (h->n_total + 1) * rec * 8) >> 20) + " MiB)";
      return 2;
    }
    h->have_traj = true;
    h->ck->writer = h;
    h->records = mode == kCkRecords;
    h->dense = mode == kCkStages;
    h->segments = mode == kCkSegments;
  }
  build_segments(h);
  h->seg_first.assign(std::max(0, Tn - 1), 0); h->seg_last.assign(std::max(0, Tn - 1), -1);
  for (int si = (int)h->segs.size() - 1; si >= 0; --si) h->seg_first[h->segs[si].interval] = si;
  for (int si = 0; si < (int)h->segs.size(); ++si) h->seg_last[h->segs[si].interval] = si;
  HIP_OK(h->d_segs.ensure(std::max<size_t>(1, h->segs.size())));
  if (!h->segs.empty())
    HIP_OK(hipMemcpyAsync(h->d_segs.p, h->segs.data(), sizeof(Seg) * h->segs.size(), hipMemcpyHostToDevice, h->stream));
  std::vector<int> cursors(2 + kMaxGroups, -1);   // [0] unused, [1] non-finite flag (adaptive solves; fixed grids: pinned, below), [2+g] segment cursor of group g
  cursors[1] = 0;
  // the non-finite flag of a fixed-grid solve lives in pinned host memory: k_snapshot stores into it directly (rare, any writer wins)
  // and the host reads it after its wait -- no device-to-host copy on the stream between the forward pass and whatever follows it
  HIP_OK(h->flag_stage.ensure(64));
  int* const bad_flag = reinterpret_cast<int*>(h->flag_stage.p);
  *bad_flag = 0;
  *persist_give_up_word(h) = 0;
  HIP_OK(hipMemcpyAsync(h->d_seg_idx.p, cursors.data(), cursors.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
  if (state0) {       // through the pinned staging area (pageable DMA is slow here); NULL = every member starts at rest
    HIP_OK(h->stage.ensure(sizeof(double) * B * nb * 6));
    memcpy(h->stage.p, state0, sizeof(double) * B * nb * 6);
    HIP_OK(hipMemcpyAsync(h->d_state0.p, h->stage.p, sizeof(double) * B * nb * 6, hipMemcpyHostToDevice, h->stream));
  } else {
    HIP_OK(hipMemsetAsync(h->d_state0.p, 0, sizeof(double) * B * nb * 6, h->stream));
  }
  DevCtx c = make_ctx(h);
  if (h->segments) { c.traj = nullptr; c.rps = 1; }        // segments level: the forward pass keeps nothing but its outputs
  if (use_fn_table(h)) c.fn_tab = h->d_fn_tab.p;
  pair_plan(h, c);
  h->lig_fwd_used = !h->pair_fwd && lig_fwd_ok(h, c, 0);
  persist_plan(h, c);
  h->launches = 0;
  const bool timing = getenv("DFX_TIMING") != nullptr;
  auto since = [&](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a).count(); };
  if (timing) fprintf(stderr, "[dfx] forward: host setup before the first launch %.0f us\n", since(tw0));
  const auto tl0 = std::chrono::steady_clock::now();
  hipLaunchKernelGGL(k_init, slot_grid(h), dim3(kThreads), 0, h->stream, c, (const double*)h->d_state0.p, timepoints[0], 0, 0LL, 0LL);
  if (c.traj)
    hipLaunchKernelGGL(k_checkpoint0, dim3((unsigned)((rec + kThreads - 1) / kThreads), (unsigned)B), dim3(kThreads), 0, h->stream, c, 0LL);
  dim3 g3((unsigned)((nb * 3 + kThreads - 1) / kThreads), (unsigned)B);
  hipLaunchKernelGGL(k_snapshot, g3, dim3(kThreads), 0, h->stream, c, h->d_fields.p, 0, bad_flag, 0, 0LL);
  HIP_OK(hipEventRecord(h->ev0, h->stream));
  if (fork_groups(h)) return 2;
  const bool eager = solve_is_eager(h) || h->segments || h->persist_fwd;      // (a persistent segment is three launches: nothing to replay)
  for (size_t si = 0; si < h->segs.size(); ++si) {
    const Seg& sg = h->segs[si];
    if (eager) enqueue_interleaved(h, c, sg.n_steps, 0);
    for (int gi = 0; gi < (int)h->groups.size(); ++gi) {
      if (!eager) if (int rc = run_segment(h, c, gi, sg.n_steps, 0)) return rc;
      if (sg.j0 + sg.n_steps == h->spis[sg.interval]) {   // buffer 0 holds the state at the end of the interval
        const Group& gr = h->groups[gi];
        // end of the interval: the state is in buffer 0, or (records checkpoint) only in the trajectory
        hipLaunchKernelGGL(k_snapshot, dim3(g3.x, gr.nm), dim3(kThreads), 0, gr.stream, group_ctx(h, c, gi), h->d_fields.p, sg.interval + 1,
                           bad_flag, c.rps > 1 ? -1 : (h->pair_fwd ? state_buf(h->step0[sg.interval + 1]) : 0),
                           (long long)h->step0[sg.interval + 1]);
      }
    }
  }
  if (join_groups(h)) return 2;
  HIP_OK(hipEventRecord(h->ev1, h->stream));
  if (timing) fprintf(stderr, "[dfx] forward: launches enqueued in %.0f us\n", since(tl0));
  if (fields) HIP_OK(hipMemcpyAsync(fields, h->d_fields.p, sizeof(double) * B * Tn * nb * 6, hipMemcpyDeviceToHost, h->stream));
  h->have_fields = true;
  memset(&h->fwd_stats, 0, sizeof(h->fwd_stats));
  h->fwd_stats.steps = h->n_total;
  h->fwd_stats.rhs_evals = h->n_total * pl.tab.s;
  h->fwd_stats.launches = h->launches;
  h->fwd_stats.streams = (int64_t)h->groups.size();
  h->fwd_stats.stage_checkpoint = c.AD ? 1 : 0;
  h->fwd_stats.checkpoint_records = h->segments ? 2 : (c.rps > 1 ? 1 : 0);
  h->fwd_stats.tile_kernels = h->persist_fwd ? 3 : kernel_build_code(h, c, h->lig_fwd_used);
  if (h->defer_forward_sync) return 0;          // the fused call goes on enqueueing the reverse sweep; finish_forward after its last wait
  HIP_OK(hipStreamSynchronize(h->stream));
  HIP_OK(hipGetLastError());
  if (timing) fprintf(stderr, "[dfx] forward: all done %.0f us after the first launch\n", since(tl0));
  return finish_forward(h, stats);
}


// ---- adaptive forward (reference odeint semantics) ------------------------------------------------
int dfx_forward_adaptive(dfx_handle* h, const double* state0, const double* timepoints, int32_t n_timepoints,
                         double rtol, double atol, int64_t max_attempts, double* fields, dfx_stats* stats) {
  HIP_OK(hipSetDevice(h->device));
  h->persist_fwd = false;
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
  HIP_OK(h->d_err_partial.ensure(B * n_wg * kWavesPerWg));
  h->n_counts = std::max(0, Tn - 1);
  HIP_OK(h->d_step_counts.ensure(std::max<size_t>(1, B * h->n_counts)));
  HIP_OK(hipMemsetAsync(h->d_step_counts.p, 0, sizeof(int) * std::max<size_t>(1, B * h->n_counts), h->stream));
  HIP_OK(h->d_acc_times.ensure(B * (size_t)kAccCap));
  HIP_OK(h->d_ts.ensure(Tn));
  HIP_OK(h->d_tmp.ensure(std::max<size_t>(B * nb * 6, B)));
  HIP_OK(hipMemcpyAsync(h->d_ts.p, timepoints, sizeof(double) * Tn, hipMemcpyHostToDevice, h->stream));
  std::vector<double> rest;
  if (!state0) { rest.assign(B * nb * 6, 0.0); state0 = rest.data(); }      // NULL = every member starts at rest, as in dfx_forward
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
  hipLaunchKernelGGL(k_init, slot_grid(h), dim3(kThreads), 0, h->stream, c, (const double*)h->d_state0.p, timepoints[0], 0, 0LL, 0LL);
  hipLaunchKernelGGL(k_snapshot, g3, dim3(kThreads), 0, h->stream, c, h->d_fields.p, 0, h->d_seg_idx.p + 1, 0, 0LL);
  launch_fwd(h, c, 0, 0, 0, -1, 0, 0);                      // A_0 = f(y0, t0)
  // only the evaluation just made comes back (row 0 / row 1 of every member's seven stage accelerations), not all of d_A
  std::vector<double> A0((size_t)B * nd), A1((size_t)B * nd);
  HIP_OK(hipMemcpy2DAsync(A0.data(), sizeof(double) * nd, h->d_A.p, sizeof(double) * 7 * nd, sizeof(double) * nd, B,
                          hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  // initial step size (Hairer II.4 as restated by jax, order 4), per member
  std::vector<double> y1(B * 2 * nd), tm(B), h0(B), d1v(B);
  for (size_t m = 0; m < B; ++m) {
    const double* q = state0 + m * 2 * nd; const double* v = q + nd; const double* a = A0.data() + m * nd;
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
  HIP_OK(hipMemcpy2DAsync(A1.data(), sizeof(double) * nd, h->d_A.p + nd, sizeof(double) * 7 * nd, sizeof(double) * nd, B,
                          hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  d_tm.release();
  for (size_t m = 0; m < B; ++m) {
    const double* q = state0 + m * 2 * nd; const double* v = q + nd;
    const double* a0 = A0.data() + m * nd; const double* a1 = A1.data() + m * nd; const double* v1 = y1.data() + m * 2 * nd + nd;
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
  // one of the eight launches of an attempt (p = 0..4: evaluations at S_1..S_5, the last one leaves the candidate y1 in buffer 3;
  // 5: the FSAL evaluation with the error estimate; 6: controller; 7: dense output / commit / next stage-1 record) for the members of
  // one context (the whole batch, or one member group on its own stream)
  auto launch_phase = [&](int p, const DevCtx& cc, hipStream_t st, dim3 grid, unsigned nm) {
    static const int inb[6] = {0, 1, 2, 1, 2, 1}, outb[6] = {0, 2, 1, 2, 1, 3};
    if (p < 5) { launch_fwd(h, cc, st, grid, p + 1, 0, inb[p + 1], outb[p + 1], 0, 0); return; }
    if (p == 5) {
#define DFX_ERR_CASE(M) case M: if (pl.contact == 2) hipLaunchKernelGGL((k_fwd_stage<M, 2>), grid, dim3(kThreads), 0, st, cc, sc_err, 6, 0, 3, -1, 0, 2); else if (pl.contact) hipLaunchKernelGGL((k_fwd_stage<M, 1>), grid, dim3(kThreads), 0, st, cc, sc_err, 6, 0, 3, -1, 0, 2); else hipLaunchKernelGGL((k_fwd_stage<M, 0>), grid, dim3(kThreads), 0, st, cc, sc_err, 6, 0, 3, -1, 0, 2); break;
      switch (pl.model) { DFX_ERR_CASE(kNonlinear) DFX_ERR_CASE(kLinearized) DFX_ERR_CASE(kSimpleSpring) DFX_ERR_CASE(kStretchTorsion) }
#undef DFX_ERR_CASE
    } else if (p == 6) hipLaunchKernelGGL(k_control, dim3(nm), dim3(kThreads), 0, st, cc, n_partials, 2.0 * (double)n_free, Tn);
    else hipLaunchKernelGGL(k_prepare, grid, dim3(kThreads), 0, st, cc, dc, Tn);
    h->launches++;
  };
  auto enqueue_attempt = [&]() { for (int p = 0; p < 8; ++p) launch_phase(p, c, h->stream, slot_grid(h), (unsigned)B); };
  HIP_OK(hipEventRecord(h->ev0, h->stream));
  // stage-1 record of the first attempt (accept = 0: nothing to commit)
  hipLaunchKernelGGL(k_prepare, slot_grid(h), dim3(kThreads), 0, h->stream, c, dc, Tn);
  const int kAttemptsPerGraph = 32;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  // Same rule as the fixed grid (solve_is_eager): launches that fill the chip are issued eagerly -- instantiating the 256-node graph
  // cost 5-10 ms per call, more than a short solve runs (profiles/r02_adaptive_fixed_cost.txt); small lattices replay a graph,
  // kept in the handle while the arguments baked into it stay the same.
  const long long waves = (long long)pl.batch * ((pl.n_slots + 63) / 64);
  if (h->use_graph && waves < 2048) {
    dfx_handle::AdaptiveKey key;
    memset(&key, 0, sizeof(key));
    key.ctx = c; key.n_timepoints = Tn; key.n_partials = n_partials; key.two_n_free = 2.0 * (double)n_free;
    if (h->adaptive_exec && memcmp(&key, &h->adaptive_key, sizeof(key)) != 0) {
      (void)hipGraphExecDestroy(h->adaptive_exec); h->adaptive_exec = nullptr;
    }
    if (!h->adaptive_exec) {
      const long long before = h->launches;
      HIP_OK(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
      for (int a = 0; a < kAttemptsPerGraph; ++a) enqueue_attempt();
      hipError_t ce = hipStreamEndCapture(h->stream, &graph);
      h->launches = before;
      if (ce == hipSuccess) ce = hipGraphInstantiate(&h->adaptive_exec, graph, nullptr, nullptr, 0);
      if (graph) (void)hipGraphDestroy(graph);
      if (ce != hipSuccess) {
        h->adaptive_exec = nullptr;
        h->err = std::string("graph capture / instantiate: ") + hipGetErrorString(ce); h->adaptive = false; return 2;
      }
      memcpy(&h->adaptive_key, &key, sizeof(key));
    }
    exec = h->adaptive_exec;
  }
  long long attempts_issued = 0;
  int rc = 0;
  while (true) {
    if (exec) { HIP_OK(hipGraphLaunch(exec, h->stream)); h->launches += 8LL * kAttemptsPerGraph; }
    else if (h->groups.size() > 1) {
      // eager launches that fill the chip: the member groups advance on their own streams, interleaved launch by launch like the
      // fixed grid (every member carries its own clock, so the groups are independent); the main stream waits for all of them
      // before the clocks are read
      if (int rc2 = fork_groups(h)) return rc2;
      for (int a = 0; a < kAttemptsPerGraph; ++a)
        for (int p = 0; p < 8; ++p)
          for (int gi = 0; gi < (int)h->groups.size(); ++gi) {
            const Group& gr = h->groups[gi];
            launch_phase(p, group_ctx(h, c, gi), gr.stream, slot_grid(h, gr), (unsigned)gr.nm);
          }
      if (int rc2 = join_groups(h)) return rc2;
    }
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
    stats->streams = (!exec && h->groups.size() > 1) ? (int64_t)h->groups.size() : 1;
    stats->stage_kernel_us = att ? 1e3 * ms / (double)(att * 8) : 0.0;
  }
  return 0;
}

// reverse sweep with the output cotangents already in h->d_G
// accumulators_cleared: the caller's prelude launch has already zeroed the gradient accumulators and set the cursors (adjoint_kinetic
// does it in the launch that clears its cotangents) -- an argument, not handle state: a flag left behind by a call that failed half way
// made the next sweep skip its zeroing (round-4 advice)
static int run_adjoint(dfx_handle* h, const dfx_grads* want, dfx_grads* grads, dfx_grads* views, dfx_stats* stats, bool kinetic, int n_target,
                       bool accumulators_cleared = false) {
  const Plan& pl = h->pl;
  const size_t B = pl.batch;
  const int Tn = (int)h->ts.size();
  set_grad_wishes(h, want);
  DevCtx c = make_ctx(h);
  h->launches = 0;
  const bool timing = getenv("DFX_TIMING") != nullptr;
  const auto ta0 = std::chrono::steady_clock::now();
  auto since = [&](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a).count(); };
  const int nseg = (int)h->segs.size();
  if (!accumulators_cleared && zero_grad_accumulators(h, nullptr, 0, nseg)) return 2;
  const double h_last = Tn > 1 ? (h->t_steps.empty() ? (h->ts[Tn - 1] - h->ts[Tn - 2]) / h->spis[Tn - 2]
                                                     : h->t_steps[h->n_total] - h->t_steps[h->n_total - 1]) : 0.0;
  HIP_OK(hipEventRecord(h->ev2, h->stream));
  pair_plan(h, c);
  if (use_fn_table(h)) c.fn_tab = h->d_fn_tab.p;
  // tile kernels: their accumulators are ligament-major (decided here, not in the launch functions: a graph replay does not call them)
  h->lig_used = h->lig_adj_used = !h->pair_adj && lig_adj_ok(h, c, -1, 0);
  persist_plan_adj(h, c);
  // the (w, Kbar_q) buffers alternate per launch, lambda (pair launches only) per step
  const int wb = (int)((h->n_total * step_units(h, 1) - 1) & 1);
  hipLaunchKernelGGL(k_adj_begin, slot_grid(h), dim3(kThreads), 0, h->stream, c, h_last, pl.tab.a[pl.tab.s][pl.tab.s - 1], wb,
                     h->pair_adj ? (int)(h->n_total & 1) : 0, (long long)h->n_total);
  if (c.AD && h->n_total > 0) {     // stage checkpoint: the record the first reverse launch reads
    const long long nr = h->n_total - 1;
    const double t_nr = h->t_steps.empty() ? h->ts[Tn - 1] - h_last : h->t_steps[nr];
    hipLaunchKernelGGL(k_rebuild_first, slot_grid(h), dim3(kThreads), 0, h->stream, c, stage_coef(pl.tab, pl.tab.s - 2), pl.tab.s - 1, nr, h_last, t_nr);
    h->launches++;
  }
  if (fork_groups(h)) return 2;
  const bool eager = solve_is_eager(h) || h->persist_adj;
  if (h->segments) {
    // output intervals backwards: records of interval k rebuilt by re-running its forward pass from the resident output row k
    // (bit-identical to the first pass: same state, same arithmetic), then its reverse stages read them
    const size_t nb6 = (size_t)pl.n_blocks * 6;
    for (int k = Tn - 2; k >= 0; --k) {
      // the buffer holds the records of THIS interval only: shift the base so that the kernels keep indexing by the global step
      // (a per-interval offset inside the kernels cost the forward kernel two hot-path spills: profiles/r02_fwd_spill_regression.txt)
      c.traj = h->ck->traj.p - (size_t)h->step0[k] * (size_t)c.rps * pl.batch * ((size_t)pl.n_blocks * kStep);
      for (int gi = 0; gi < (int)h->groups.size(); ++gi) {
        const Group& gr = h->groups[gi];
        const DevCtx cg = group_ctx(h, c, gi);
        hipLaunchKernelGGL(k_init, slot_grid(h, gr), dim3(kThreads), 0, gr.stream, cg, (const double*)(h->d_fields.p + (size_t)k * nb6), h->ts[k], 0,
                           (long long)((size_t)Tn * nb6), (long long)h->step0[k]);
        hipLaunchKernelGGL(k_checkpoint0, dim3((unsigned)(((size_t)pl.n_blocks * kStep + kThreads - 1) / kThreads), (unsigned)gr.nm), dim3(kThreads), 0,
                           gr.stream, cg, (long long)h->step0[k]);
        h->launches += 2;
      }
      for (int si = h->seg_first[k]; si <= h->seg_last[k]; ++si) enqueue_interleaved(h, c, h->segs[si].n_steps, 0, si);
      for (int si = h->seg_last[k]; si >= h->seg_first[k]; --si) enqueue_interleaved(h, c, h->segs[si].n_steps, 1, si);
    }
  } else
  for (int si = nseg - 1; si >= 0; --si) {
    if (eager) { enqueue_interleaved(h, c, h->segs[si].n_steps, 1); continue; }
    for (int gi = 0; gi < (int)h->groups.size(); ++gi)
      if (int rc = run_segment(h, c, gi, h->segs[si].n_steps, 1)) return rc;
  }
  if (join_groups(h)) return 2;
  if (kinetic) {
    dim3 g((unsigned)((n_target * 3 + 63) / 64), (unsigned)B);
    hipLaunchKernelGGL(k_kinetic_mass_grad, g, dim3(64), 0, h->stream, c, (const double*)h->d_fields.p, (const int32_t*)h->d_target.p, n_target);
  }
  HIP_OK(hipEventRecord(h->ev3, h->stream));
  if (timing) fprintf(stderr, "[dfx] adjoint: sweep enqueued %.0f us after entry\n", since(ta0));
  if (int rc = collect_grads(h, want, grads, views, true)) return rc;
  if (*persist_give_up_word(h)) { h->err = std::string("adjoint: ") + kPersistGaveUp; return 2; }
  if (timing) {
    float ms0 = 0.f;
    (void)hipEventElapsedTime(&ms0, h->ev2, h->ev3);
    fprintf(stderr, "[dfx] adjoint: gradients collected %.0f us after entry (sweep on the device: %.0f us)\n", since(ta0), 1e3 * ms0);
  }
  if (stats) {
    memset(stats, 0, sizeof(*stats));
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, h->ev2, h->ev3);
    stats->steps = h->n_total;
    stats->rhs_evals = h->n_total * pl.tab.s;
    stats->launches = h->launches;
    stats->kernel_ms = ms;
    stats->streams = (int64_t)h->groups.size();
    stats->stage_kernel_us = h->n_total ? 1e3 * ms / (double)(h->n_total * pl.tab.s * 2) : 0.0;
    stats->stage_checkpoint = c.AD ? 1 : 0;
    stats->checkpoint_records = h->segments ? 2 : (c.rps > 1 ? 1 : 0);
    stats->tile_kernels = h->persist_adj ? 3 : ((c.g_b || c.AD) ? 0 : kernel_build_code(h, c, h->lig_adj_used));
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

static const char* kStaleCheckpoint = "the shared trajectory checkpoint was overwritten by a solve of another handle (dfx_share_checkpoint): run this handle's forward again";

int dfx_adjoint(dfx_handle* h, const double* fields_bar, dfx_grads* grads, dfx_stats* stats) {
  HIP_OK(hipSetDevice(h->device));
  if (!h->have_traj) { h->err = "adjoint: run forward with keep_trajectory=1 first"; return 1; }
  if (h->ck->writer != h) { h->err = std::string("adjoint: ") + kStaleCheckpoint; return 1; }
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
  return run_adjoint(h, grads, grads, nullptr, stats, false, 0);
}

static int upload_targets(dfx_handle* h, const int32_t* target_blocks, int32_t n_target, bool copy = true) {
  for (int i = 0; i < n_target; ++i)
    if (target_blocks[i] < 0 || target_blocks[i] >= h->pl.n_blocks) { h->err = "target block out of range"; return 1; }
  HIP_OK(h->d_target.ensure(std::max(1, n_target)));
  if (copy) HIP_OK(hipMemcpyAsync(h->d_target.p, target_blocks, sizeof(int32_t) * n_target, hipMemcpyHostToDevice, h->stream));
  HIP_OK(h->d_obj.ensure(h->pl.batch));
  return 0;
}

int dfx_objective_kinetic(dfx_handle* h, const int32_t* target_blocks, int32_t n_target, double* objective) {
  HIP_OK(hipSetDevice(h->device));
  if (!h->have_fields) { h->err = "objective: run forward first"; return 1; }
  if (int rc = upload_targets(h, target_blocks, n_target)) return rc;
  DevCtx c = make_ctx(h);
  hipLaunchKernelGGL(k_kinetic, dim3(h->pl.batch), dim3(kThreads), 0, h->stream, c, (const double*)h->d_fields.p,
                     (const int32_t*)h->d_target.p, n_target, (double*)nullptr, h->d_obj.p, (double*)nullptr);
  HIP_OK(hipMemcpyAsync(objective, h->d_obj.p, sizeof(double) * h->pl.batch, hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  return 0;
}

static int adjoint_kinetic(dfx_handle* h, const int32_t* target_blocks, int32_t n_target, double* objective, const dfx_grads* want,
                           dfx_grads* grads, dfx_grads* views, dfx_stats* stats) {
  HIP_OK(hipSetDevice(h->device));
  if (!h->have_traj || !h->have_fields) { h->err = "adjoint_kinetic: run forward with keep_trajectory=1 first"; return 1; }
  if (h->ck->writer != h) { h->err = std::string("adjoint_kinetic: ") + kStaleCheckpoint; return 1; }
  const Plan& pl = h->pl;
  const size_t B = pl.batch, nb = pl.n_blocks;
  const int Tn = (int)h->ts.size();
  if (ensure_adjoint_buffers(h)) return 2;
  if (int rc = upload_targets(h, target_blocks, n_target, false)) return rc;
  HIP_OK(h->d_G.ensure(B * Tn * nb * 6));
  // one launch: accumulators and cotangents cleared, cursors at the last segment, target blocks in place (zero_grad_accumulators)
  set_grad_wishes(h, want);
  if (zero_grad_accumulators(h, h->d_G.p, B * Tn * nb * 6, (int)h->segs.size(), target_blocks, n_target)) return 2;
  DevCtx c = make_ctx(h);
  // the objective rides along with the reverse sweep: the kernel stores it into pinned host memory as well (a copy on the stream would be
  // a hop to the copy engine and back in front of the sweep), read after the sweep's final synchronisation
  if (objective) HIP_OK(h->obj_stage.ensure(sizeof(double) * B));
  hipLaunchKernelGGL(k_kinetic, dim3(h->pl.batch), dim3(kThreads), 0, h->stream, c, (const double*)h->d_fields.p,
                     (const int32_t*)h->d_target.p, n_target, h->d_G.p, h->d_obj.p, objective ? reinterpret_cast<double*>(h->obj_stage.p) : (double*)nullptr);
  if (int rc = run_adjoint(h, want, grads, views, stats, true, n_target, true)) return rc;
  if (objective) memcpy(objective, h->obj_stage.p, sizeof(double) * B);
  return 0;
}

int dfx_adjoint_kinetic(dfx_handle* h, const int32_t* target_blocks, int32_t n_target, dfx_grads* grads, dfx_stats* stats) {
  return adjoint_kinetic(h, target_blocks, n_target, nullptr, grads, grads, nullptr, stats);
}

int dfx_kinetic_value_and_grad(dfx_handle* h, const int32_t* target_blocks, int32_t n_target, double* objective,
                               const dfx_grads* want, dfx_grads* views, dfx_stats* stats) {
  return adjoint_kinetic(h, target_blocks, n_target, objective, want, nullptr, views, stats);
}

int dfx_kinetic_value_and_grad_device(dfx_handle* h, const int32_t* target_blocks, int32_t n_target, double* objective,
                                      const dfx_grads* want, dfx_grads* device_views, dfx_stats* stats) {
  h->device_views = true;
  const int rc = adjoint_kinetic(h, target_blocks, n_target, objective, want, nullptr, device_views, stats);
  h->device_views = false;
  return rc;
}

int dfx_forward_kinetic_value_and_grad(dfx_handle* h, const double* state0, const double* timepoints, int32_t n_timepoints,
                                       const int32_t* steps_per_interval, const int32_t* target_blocks, int32_t n_target,
                                       double* objective, const dfx_grads* want, dfx_grads* views, int32_t device_views,
                                       dfx_stats* forward_stats, dfx_stats* adjoint_stats) {
  h->defer_forward_sync = true;
  int rc = forward_grid_impl(h, state0, timepoints, n_timepoints, steps_per_interval, nullptr, 1, nullptr, nullptr, false);
  h->defer_forward_sync = false;
  if (rc) return rc;
  h->device_views = device_views != 0;
  rc = adjoint_kinetic(h, target_blocks, n_target, objective, want, nullptr, views, adjoint_stats);
  h->device_views = false;
  HIP_OK(hipStreamSynchronize(h->stream));          // (already idle when the sweep returned normally)
  const int rcf = finish_forward(h, forward_stats);
  return rcf ? rcf : rc;
}

// device -> caller memory through the pinned staging area, in chunks (outputs here can be GBs; pageable DMA is slow)
static int download(dfx_handle* h, double* dst, const double* src, size_t n) {
  const size_t chunk = (size_t)8 << 20;      // doubles per chunk: 64 MiB
  HIP_OK(h->stage.ensure(std::min(n, chunk) * sizeof(double)));
  for (size_t off = 0; off < n; off += chunk) {
    const size_t cnt = std::min(chunk, n - off);
    HIP_OK(hipMemcpyAsync(h->stage.p, src + off, cnt * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_OK(hipStreamSynchronize(h->stream));
    memcpy(dst + off, h->stage.p, cnt * sizeof(double));
  }
  return 0;
}

int dfx_download(dfx_handle* h, double* dst, const double* device_src, int64_t n) {
  HIP_OK(hipSetDevice(h->device));
  if (n < 0 || (n && (!dst || !device_src))) { h->err = "dfx_download: bad arguments"; return 1; }
  return n ? download(h, dst, device_src, (size_t)n) : 0;
}

int dfx_response_data(dfx_handle* h, double* strain_energy_stretch, double* strain_energy_shear, double* strain_energy_bending,
                      double* kinetic_energy) {
  HIP_OK(hipSetDevice(h->device));
  if (!h->have_fields || !h->have_params) { h->err = "response_data: run forward first"; return 1; }
  const Plan& pl = h->pl;
  const size_t B = pl.batch, nb = pl.n_blocks, nbd = pl.n_bonds, T = h->ts.size();
  const bool bonds = strain_energy_stretch || strain_energy_shear || strain_energy_bending;
  HIP_OK(h->d_resp.ensure((bonds ? 3 * B * T * nbd : 0) + (kinetic_energy ? B * T * nb : 0) + 1));
  double* d_s = h->d_resp.p;
  double* d_sh = d_s + (bonds ? B * T * nbd : 0);
  double* d_b = d_sh + (bonds ? B * T * nbd : 0);
  double* d_k = d_b + (bonds ? B * T * nbd : 0);
  DevCtx c = make_ctx(h);
  dim3 grid((unsigned)((pl.n_slots + kThreads - 1) / kThreads), (unsigned)T, (unsigned)B);
  hipLaunchKernelGGL(k_response, grid, dim3(kThreads), 0, h->stream, c, (const double*)h->d_fields.p, (const int32_t*)h->d_slot_bond.p, (const int32_t*)h->d_ovf_bond.p, (int)nbd,
                     bonds ? d_s : (double*)nullptr, bonds ? d_sh : (double*)nullptr, bonds ? d_b : (double*)nullptr,
                     kinetic_energy ? d_k : (double*)nullptr);
  HIP_OK(hipGetLastError());
  if (strain_energy_stretch) if (int rc = download(h, strain_energy_stretch, d_s, B * T * nbd)) return rc;
  if (strain_energy_shear) if (int rc = download(h, strain_energy_shear, d_sh, B * T * nbd)) return rc;
  if (strain_energy_bending) if (int rc = download(h, strain_energy_bending, d_b, B * T * nbd)) return rc;
  if (kinetic_energy) if (int rc = download(h, kinetic_energy, d_k, B * T * nb)) return rc;
  HIP_OK(hipStreamSynchronize(h->stream));
  return 0;
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
  hipLaunchKernelGGL(k_init, slot_grid(h), dim3(kThreads), 0, h->stream, c, (const double*)h->d_state0.p, t, 0, 0LL, 0LL);
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
  if (int rc = collect_grads(h, grads, grads, nullptr, false)) return rc;
  for (size_t m = 0; m < B; ++m)
    for (size_t b = 0; b < nb; ++b)
      for (int d = 0; d < 3; ++d) {
        y_bar[m * nb * 6 + b * 3 + d] = YB[m * pl.tab.s * nb * 6 + b * 6 + (c.lam_pairs ? 2 * d : d)];          // DevCtx::lam_pairs
        y_bar[m * nb * 6 + nb * 3 + b * 3 + d] = YB[m * pl.tab.s * nb * 6 + b * 6 + (c.lam_pairs ? 2 * d + 1 : 3 + d)];
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
      r[3] = sin(0.5 * r[2]);
    }
  HIP_OK(hipMemcpyAsync(h->d_POS.p, S.data(), sizeof(double) * S.size(), hipMemcpyHostToDevice, h->stream));
  HIP_OK(h->d_tmp.ensure(B * pl.n_slots));
  DevCtx c = make_ctx(h);
#define DFX_EN_CASE(M) case M: if (pl.contact == 2) hipLaunchKernelGGL((k_energy<M, 2>), slot_grid(h), dim3(kThreads), 0, h->stream, c, h->d_tmp.p); else if (pl.contact) hipLaunchKernelGGL((k_energy<M, 1>), slot_grid(h), dim3(kThreads), 0, h->stream, c, h->d_tmp.p); else hipLaunchKernelGGL((k_energy<M, 0>), slot_grid(h), dim3(kThreads), 0, h->stream, c, h->d_tmp.p); break;
  switch (pl.model) { DFX_EN_CASE(kNonlinear) DFX_EN_CASE(kLinearized) DFX_EN_CASE(kSimpleSpring) DFX_EN_CASE(kStretchTorsion) }
#undef DFX_EN_CASE
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
