// engine_launch.hip -- libdfx host side: which build of the stage kernels a launch takes, segments as stage launches / hipGraph replays / persistent
// launches, member groups
// (one of five translation units; shared declarations in dfx_engine.h, the design in DESIGN.md section 3)
#include "dfx_engine.h"

using namespace dfx_persist;

void drop_graphs(dfx_handle* h) {
  for (auto& kv : h->graphs) (void)hipGraphExecDestroy(kv.second);
  h->graphs.clear();
  h->graph_ctx_valid = false;
  if (h->adaptive_exec) { (void)hipGraphExecDestroy(h->adaptive_exec); h->adaptive_exec = nullptr; }
}

DevCtx make_ctx(dfx_handle* h) {
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
  // (general bond lists run the one reverse build that has them)
  c.g_r = h->d_g_r.p; c.g_phi = h->d_g_phi.p; c.g_b = (h->want_bond_grads || pl.n_ovf) ? h->d_g_b.p : nullptr;
  c.touch = h->d_touch.p;
  c.lam_pairs = (c.g_b || c.AD) ? 0 : 1;      // the REBUILD builds of the reverse stage (launch_adj_t) keep the scalar layout
  c.blk_m = h->d_blk_m.p; c.blk_c = h->want_damping_grads ? h->d_blk_c.p : nullptr;
  c.fn_g = h->want_fn_grads ? h->d_fn_g.p : nullptr;
  return c;
}

dim3 slot_grid(const dfx_handle* h) { return dim3((h->pl.n_slots + kThreads - 1) / kThreads, h->pl.batch); }
dim3 slot_grid(const dfx_handle* h, const Group& g) { return dim3((h->pl.n_slots + kThreads - 1) / kThreads, g.nm); }
// context of one group: same arrays, its own member range and its own segment cursor
DevCtx group_ctx(const dfx_handle* h, const DevCtx& c, int gi) {
  DevCtx cg = c;
  cg.m0 = h->groups[gi].m0;
  cg.cur = h->d_cur.p + gi;
  return cg;
}

StageCoef stage_coef(const Tableau& T, int i) {
  StageCoef sc;
  memset(&sc, 0, sizeof(sc));
  const int r = i + 1;
  for (int l = 0; l <= i; ++l) { sc.cv[l] = T.a[r][l]; sc.cq[l] = T.aa[r][l]; }
  sc.c_i = T.c[i];
  sc.c_next = T.c[r];
  return sc;
}
AdjCoef adj_coef(const Tableau& T, int i) {
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

#ifdef DFX_EXPERIMENTAL
#include "dfx_experimental_host.h"      // two stages per launch on lattice windows, every ligament once on lattice tiles: opt-in experiments
#else
void setup_tiling(dfx_handle*) {}
void setup_lig(dfx_handle*) {}
int lig_pack(dfx_handle*) { return 0; }
void pair_plan(dfx_handle* h, const DevCtx&) { h->pair_fwd = h->pair_adj = false; }
bool lig_fwd_ok(const dfx_handle*, const DevCtx&, int) { return false; }
bool lig_adj_ok(const dfx_handle*, const DevCtx&, int, int) { return false; }
#endif
// stage buffer that holds the state of step n at the end of an output interval (the pair launches ping-pong it by step parity)
int pair_state_buf(const dfx_handle* h, long long n) {
#ifdef DFX_EXPERIMENTAL
  if (h->pair_fwd) return state_buf(n);
#endif
  (void)h; (void)n;
  return 0;
}

// 3-node blocks: the two stage kernels pack five triangles per 16 lanes (lane_pos<3>) instead of leaving every fourth lane idle --
// 20 blocks per wave instead of 16.  Fixed grid only (the adaptive controller's error reduction keeps the quad mapping), not with the
// distance-based contact; reverse: the records build.  DFX_PACK3=0 keeps the quad mapping (A/B measurements).
bool pack3(const dfx_handle* h) {
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
#define DFX_FWD_I(I)                                                                                                              \
  case I:                                                                                                                         \
    if (recs)                                                                                                                     \
      hipLaunchKernelGGL((k_fwd_stage<MODEL, CONTACT, NPB, 1, 0, 1, I, 1>), grid, dim3(kThreads), 0, st, c, scf, i, j, in_buf, out_buf, \
                         y_buf, mode);                                                                                            \
    else                                                                                                                          \
      hipLaunchKernelGGL((k_fwd_stage<MODEL, CONTACT, NPB, 1, 0, 1, I, 0>), grid, dim3(kThreads), 0, st, c, scf, i, j, in_buf, out_buf, \
                         y_buf, mode);                                                                                            \
    return true;
    switch (i) { DFX_FWD_I(0) DFX_FWD_I(1) DFX_FWD_I(2) DFX_FWD_I(3) DFX_FWD_I(4) DFX_FWD_I(5) default: break; }
#undef DFX_FWD_I
  }
  return false;
}
// the reverse per-stage builds live in a translation unit of their own (stage_builds_adj.hip: compiled with its own scheduling strategy)
namespace dfx_hot {
bool launch_adj_stage_build(int model, int contact, int npb, hipStream_t st, dim3 grid, const DevCtx& c, const AdjCoef& acf, int i, int j, int in_buf, int wbuf,
                            int local_only, const StageCoef& rc, int rb);
}
template <int MODEL, int CONTACT, int NPB>
static bool launch_adj_hot(dfx_handle* h, const DevCtx& c, hipStream_t st, dim3 grid, int i, int j, int in_buf, int wbuf, int local_only, const StageCoef& rc,
    int rb) {
  if constexpr ((MODEL == kNonlinear || MODEL == kLinearized) && CONTACT != 2) {
    if (!h->wt || !h->stage_builds || !(c.k_uniform && c.l_dict_on && c.l_dict_lds && c.damping_uniform && !c.t_steps)) return false;
    return dfx_hot::launch_adj_stage_build(MODEL, CONTACT, NPB, st, grid, c, adj_coef(h->pl.tab, i), i, j, in_buf, wbuf, local_only, rc, rb);
  }
  return false;
}
// what dfx_stats.tile_kernels reports: 1 tile kernels, 2 the per-stage / common-shape builds of the slot kernels, 0 their generic builds
int kernel_build_code(const dfx_handle* h, const DevCtx& c, bool tile) {
  if (tile) return 1;
  const bool model_ok = (h->pl.model == kNonlinear || h->pl.model == kLinearized) && h->pl.contact != DFX_CONTACT_DISTANCE;
  const bool quad_or_packed = !h->pl.n_ovf && (h->pl.n_npb == 4 || pack3(h));
  return (model_ok && quad_or_packed && h->wt && h->stage_builds && c.fn_tab && hot_shape(c) && h->pl.tab.s <= 6) ? 2 : 0;
}
template <int MODEL, int CONTACT>
static void launch_fwd_t(dfx_handle* h, const DevCtx& c, hipStream_t st, dim3 grid, int i, int j, int in_buf, int out_buf, int y_buf, int mode) {
#ifdef DFX_EXPERIMENTAL
  if constexpr (CONTACT != 2) {
    if (lig_fwd_ok(h, c, mode)) {
      const dim3 tg(h->lig.n_wg, grid.y);
      hipLaunchKernelGGL((k_fwd_tile<MODEL, CONTACT>), tg, dim3(64 * kTileWaves), 0, st, c, h->lig, stage_coef(h->pl.tab, i), i, j, in_buf, out_buf,
                         y_buf, mode);
      return;
    }
  }
#endif
  if (h->pl.n_ovf) {       // general bond lists: the build that walks a node's extra ligaments (quad mapping, in-kernel time functions)
    if constexpr (CONTACT != 2)
      hipLaunchKernelGGL((k_fwd_stage<MODEL, CONTACT, 4, 0, 1>), grid, dim3(kThreads), 0, st, c, stage_coef(h->pl.tab, i), i, j, in_buf, out_buf, y_buf, mode);
    return;
  }
  const bool tab = c.fn_tab != nullptr && !c.clock;        // the segment's time-function table is there: the build that reads it
  if (CONTACT != 2 && pack3(h) && !(mode & 2)) {
    if constexpr (CONTACT != 2) {
      if (tab && launch_fwd_hot<MODEL, CONTACT, 3>(h, c, st, dim3(c.n_wg3, grid.y), i, j, in_buf, out_buf, y_buf, mode)) return;
      if (tab) hipLaunchKernelGGL((k_fwd_stage<MODEL, CONTACT, 3, 1>), dim3(c.n_wg3, grid.y), dim3(kThreads), 0, st, c, stage_coef(h->pl.tab, i), i, j,
          in_buf, out_buf, y_buf, mode);
      else hipLaunchKernelGGL((k_fwd_stage<MODEL, CONTACT, 3, 0>), dim3(c.n_wg3, grid.y), dim3(kThreads), 0, st, c, stage_coef(h->pl.tab, i), i, j, in_buf,
          out_buf, y_buf, mode);
    }
    return;
  }
  if (tab && h->wt) {
    if (launch_fwd_hot<MODEL, CONTACT, 4>(h, c, st, grid, i, j, in_buf, out_buf, y_buf, mode)) return;
    hipLaunchKernelGGL((k_fwd_stage<MODEL, CONTACT, 4, 1, 0, 1>), grid, dim3(kThreads), 0, st, c, stage_coef(h->pl.tab, i), i, j, in_buf, out_buf, y_buf, mode);
  }
  else if (tab) hipLaunchKernelGGL((k_fwd_stage<MODEL, CONTACT, 4, 1>), grid, dim3(kThreads), 0, st, c, stage_coef(h->pl.tab, i), i, j, in_buf, out_buf,
      y_buf, mode);
  else hipLaunchKernelGGL((k_fwd_stage<MODEL, CONTACT, 4, 0>), grid, dim3(kThreads), 0, st, c, stage_coef(h->pl.tab, i), i, j, in_buf, out_buf, y_buf, mode);
}
void launch_fwd(dfx_handle* h, const DevCtx& c, hipStream_t st, dim3 grid, int i, int j, int in_buf, int out_buf, int y_buf, int mode) {
  const Plan& pl = h->pl;
#define DFX_FWD_CASE(M) case M: if (pl.contact == 2) launch_fwd_t<M, 2>(h, c, st, grid, i, j, in_buf, out_buf, y_buf, mode); \
    else if (pl.contact) launch_fwd_t<M, 1>(h, c, st, grid, i, j, in_buf, out_buf, y_buf, mode); else launch_fwd_t<M, 0>(h, c, st, grid, i, j, in_buf, \
    out_buf, y_buf, mode); break;
  switch (pl.model) { DFX_FWD_CASE(kNonlinear) DFX_FWD_CASE(kLinearized) DFX_FWD_CASE(kSimpleSpring) DFX_FWD_CASE(kStretchTorsion) }
#undef DFX_FWD_CASE
  h->launches++;
}
void launch_fwd(dfx_handle* h, const DevCtx& c, int i, int j, int in_buf, int out_buf, int y_buf, int mode) {
  launch_fwd(h, c, h->stream, slot_grid(h), i, j, in_buf, out_buf, y_buf, mode);
}
template <int MODEL, int CONTACT>
static void launch_adj_t(dfx_handle* h, const DevCtx& c, hipStream_t st, dim3 grid, int i, int j, int in_buf, int wbuf, int local_only) {
  // stage checkpoint: which record this launch rebuilds for the launch after it (0: none)
  const int s = h->pl.tab.s;
  const int rb = (c.AD && !local_only) ? (i >= 2 ? i - 1 : (i == 0 ? s - 1 : 0)) : 0;
  const StageCoef rc = stage_coef(h->pl.tab, rb > 0 ? rb - 1 : 0);
#ifdef DFX_EXPERIMENTAL
  if constexpr (CONTACT != 2) {
    if (lig_adj_ok(h, c, wbuf, local_only)) {
      const dim3 tg(h->lig.n_wg, grid.y);
      hipLaunchKernelGGL((k_adj_tile<MODEL, CONTACT>), tg, dim3(64 * kTileWaves), 0, st, c, h->lig, adj_coef(h->pl.tab, i), i, j, in_buf);
      return;
    }
  }
#endif
  if (h->pl.n_ovf) {       // general bond lists: one build for every checkpoint level (per-ligament gradients on, rebuild on)
    if constexpr (CONTACT != 2)
      hipLaunchKernelGGL((k_adj_stage<MODEL, CONTACT, 1, 1, 4, 0, 1>), grid, dim3(kThreads), 0, st, c, adj_coef(h->pl.tab, i), i, j, in_buf, wbuf, local_only,
          rc, rb);
  }
  else if (c.g_b) hipLaunchKernelGGL((k_adj_stage<MODEL, CONTACT, 1, 1>), grid, dim3(kThreads), 0, st, c, adj_coef(h->pl.tab, i), i, j, in_buf, wbuf,
      local_only, rc, rb);
  else if (c.AD) hipLaunchKernelGGL((k_adj_stage_rb<MODEL, CONTACT>), grid, dim3(kThreads), 0, st, c, adj_coef(h->pl.tab, i), i, j, in_buf, wbuf, local_only,
      rc, rb);
  else if (CONTACT != 2 && pack3(h)) {
    if constexpr (CONTACT != 2) {
      if (c.fn_tab && !local_only && launch_adj_hot<MODEL, CONTACT, 3>(h, c, st, dim3(c.n_wg3, grid.y), i, j, in_buf, wbuf, local_only, rc, rb)) return;
      if (c.fn_tab && !local_only) hipLaunchKernelGGL((k_adj_stage<MODEL, CONTACT, 0, 0, 3, 1>), dim3(c.n_wg3, grid.y), dim3(kThreads), 0, st, c,
          adj_coef(h->pl.tab, i), i, j, in_buf, wbuf, local_only, rc, rb);
      else hipLaunchKernelGGL((k_adj_stage<MODEL, CONTACT, 0, 0, 3, 0>), dim3(c.n_wg3, grid.y), dim3(kThreads), 0, st, c, adj_coef(h->pl.tab, i), i, j,
          in_buf, wbuf, local_only, rc, rb);
    }
  }
  else if (c.fn_tab && !local_only && h->wt) {
    if (launch_adj_hot<MODEL, CONTACT, 4>(h, c, st, grid, i, j, in_buf, wbuf, local_only, rc, rb)) return;
    hipLaunchKernelGGL((k_adj_stage<MODEL, CONTACT, 0, 0, 4, 1, 0, 1>), grid, dim3(kThreads), 0, st, c, adj_coef(h->pl.tab, i), i, j, in_buf, wbuf,
        local_only, rc, rb);
  }
  else if (c.fn_tab && !local_only) hipLaunchKernelGGL((k_adj_stage<MODEL, CONTACT, 0, 0, 4, 1>), grid, dim3(kThreads), 0, st, c, adj_coef(h->pl.tab, i), i,
      j, in_buf, wbuf, local_only, rc, rb);
  else hipLaunchKernelGGL((k_adj_stage<MODEL, CONTACT, 0, 0, 4, 0>), grid, dim3(kThreads), 0, st, c, adj_coef(h->pl.tab, i), i, j, in_buf, wbuf, local_only,
      rc, rb);
}
void launch_adj(dfx_handle* h, const DevCtx& c, hipStream_t st, dim3 grid, int i, int j, int in_buf, int wbuf, int local_only) {
  const Plan& pl = h->pl;
#define DFX_ADJ_CASE(M) case M: if (pl.contact == 2) launch_adj_t<M, 2>(h, c, st, grid, i, j, in_buf, wbuf, local_only); \
    else if (pl.contact) launch_adj_t<M, 1>(h, c, st, grid, i, j, in_buf, wbuf, local_only); else launch_adj_t<M, 0>(h, c, st, grid, i, j, in_buf, wbuf, \
    local_only); break;
  switch (pl.model) { DFX_ADJ_CASE(kNonlinear) DFX_ADJ_CASE(kLinearized) DFX_ADJ_CASE(kSimpleSpring) DFX_ADJ_CASE(kStretchTorsion) }
#undef DFX_ADJ_CASE
  h->launches++;
}
void launch_adj(dfx_handle* h, const DevCtx& c, int i, int j, int in_buf, int wbuf, int local_only) {
  launch_adj(h, c, h->stream, slot_grid(h), i, j, in_buf, wbuf, local_only);
}

// ---- the stage loop without kernel boundaries (dfx_persist.h) ---------------------------------------------------------------------
// A persistent launch needs ALL its workgroups resident at once; two of them that are each half resident would wait for each other
// until their spins give up.  Residency is a matter of registers (256-thread workgroups = one wave per SIMD; the dispatcher spreads
// them evenly by itself: profiles/r05_persistent_stage_mock.txt, run 5), so the process keeps account in SLOTS of 8 registers per lane
// (the allocation granule): a SIMD has kPersistSlots = 512 / 8 of them; a workgroup takes what its kernel allocates (forward kernels
// 104-112 registers: four workgroups per compute unit, reverse kernels 168: three); a launch takes  workgroups per compute unit x slots
// per workgroup.  Launches on ONE stream follow each other anyway; launches on
// different streams (the engines of a multi-input objective, one host thread each) may overlap while the streams' largest
// outstanding needs sum to <= kPersistSlots -- otherwise the new launch first waits for everything another stream has queued.
static const int kPersistSlots = 64;
struct PersistInflight { hipEvent_t ev; hipStream_t st; int slots; };
static std::mutex g_persist_mu;
static std::vector<PersistInflight> g_persist_inflight[64];
static std::vector<hipEvent_t> g_persist_events[64];        // recycled

// slots one workgroup of `fn` takes (by its register allocation); 0: the kernel cannot be launched
static int persist_wg_slots(const void* fn) {
  static std::mutex mu;
  static std::map<const void*, int> cache;
  std::lock_guard<std::mutex> lk(mu);
  auto it = cache.find(fn);
  if (it != cache.end()) return it->second;
  hipFuncAttributes at;
  if (!fn || hipFuncGetAttributes(&at, fn) != hipSuccess) { (void)hipGetLastError(); return 0; }
  const int alloc = std::max(8, ((at.numRegs + 7) / 8) * 8);
  return cache[fn] = alloc > 512 ? 0 : alloc / 8;
}
// workgroups of `fn` a compute unit really holds at once: the runtime's own answer (registers, the 34.8 KB of LDS of the reverse loop,
// scratch -- round-5 advice: the register count alone ignores the latter two); 0: unknown
static int persist_wg_per_cu(const void* fn) {
  static std::mutex mu;
  static std::map<const void*, int> cache;
  std::lock_guard<std::mutex> lk(mu);
  auto it = cache.find(fn);
  if (it != cache.end()) return it->second;
  int nb = 0;
  if (!fn || hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, kPersistThreads, 0) != hipSuccess) { (void)hipGetLastError(); nb = 0; }
  return cache[fn] = nb;
}
bool persist_shape_ok(const dfx_handle* h) {
  const Plan& pl = h->pl;
  const char* e = getenv("DFX_PERSIST");
  if ((e && e[0] == '0') || h->persist_off) return false;
  return (pl.model == kNonlinear || pl.model == kLinearized) && pl.contact != DFX_CONTACT_DISTANCE && !pl.n_ovf && pl.tab.s <= kPersistStages &&
         (pl.n_npb == 3 || pl.n_npb == 4);
}
static int persist_waves_per_member(const dfx_handle* h, int npb) {
  return npb == 3 ? (h->pl.n_blocks + 19) / 20 : (h->pl.n_slots + 63) / 64;
}
static int persist_cap(const void* fn);
// how many members fit on the chip at once (0: not even one), and the launch shape for `nm` of them
int persist_members_that_fit(dfx_handle* h, const void* fn, int npb) {
  if (!h->n_cu) { int v = 0; if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, h->device) != hipSuccess || v <= 0) return 0; h->n_cu = v; }
  const int cap = persist_cap(fn);
  const long long wpm = persist_waves_per_member(h, npb);
  if (cap <= 0 || wpm <= 0) return 0;
  return (int)std::min<long long>(h->pl.batch, ((long long)cap * h->n_cu * 4) / wpm);
}
// workgroups of `fn` a compute unit may hold in a persistent launch (registers, the runtime's occupancy, DFX_PERSIST_MAX_WG)
static int persist_cap(const void* fn) {
  const int wg_slots = persist_wg_slots(fn);
  int cap = wg_slots ? std::min(8, kPersistSlots / wg_slots) : 0;      // (8 waves per SIMD at most)
  if (const int occ = persist_wg_per_cu(fn)) cap = std::min(cap, occ);
  if (const char* e = getenv("DFX_PERSIST_MAX_WG")) cap = std::min(cap, atoi(e));
  return cap;
}
// the launch shape for `nm` members.  *xcd_wg > 0: every member on one XCD (persist_wave, dfx_persist.h) -- taken where a member's workgroups fit
// the 32 compute units of an XCD one each and the members fit the launch that way; DFX_PERSIST_XCD=0 keeps the dense packing
static void persist_shape(const dfx_handle* h, int npb, int nm, int* grid, int* per_cu, const void* fn = nullptr, int* xcd_wg = nullptr) {
  const long long wpm = persist_waves_per_member(h, npb);
  if (xcd_wg) {
    *xcd_wg = 0;
    const char* e = getenv("DFX_PERSIST_XCD");
    const int wg = (int)((wpm + 3) / 4), cus = h->n_cu / 8, per_xcd = (nm + 7) / 8, cap = fn ? persist_cap(fn) : 0;
    if (!(e && e[0] == '0') && h->n_cu % 8 == 0 && wg <= cus && (long long)per_xcd * wg <= (long long)cap * cus) {
      *xcd_wg = wg;
      *grid = 8 * per_xcd * wg;
      *per_cu = std::max(1, (per_xcd * wg + cus - 1) / cus);
      return;
    }
  }
  const long long waves = (long long)nm * wpm;
  const long long g = (waves + 3) / 4;
  *grid = (int)g;
  *per_cu = (int)std::max<long long>(1, (g + h->n_cu - 1) / h->n_cu);
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
bool persist_members_ok(const dfx_handle* h, int per_launch) {
  return per_launch > 0 && (h->pl.batch + per_launch - 1) / per_launch <= persist_max_chunks();
}
// would both sweeps of a fixed-grid solve of this handle run the persistent loop?  (asked before the context exists: the checkpoint choice)
bool persist_would_serve(dfx_handle* h) {
  if (!persist_shape_ok(h) || h->adaptive || h->groups.size() != 1) return false;
  const int npb = (h->pl.n_npb == 3 && pack3(h)) ? 3 : 4;
  return persist_members_ok(h, persist_members_that_fit(h, dfx_persist::fwd_kernel(h->pl.model, h->pl.contact, npb), npb)) &&
         persist_members_ok(h, persist_members_that_fit(h, dfx_persist::adj_kernel(h->pl.model, h->pl.contact, npb), npb));
}
// decided per solve, after the context is known
void persist_plan(dfx_handle* h, const DevCtx& c) {
  h->persist_fwd = false;
  if (h->pair_fwd || h->lig_fwd_used || !persist_common_ok(h, c)) return;
  const void* fn = dfx_persist::fwd_kernel(h->pl.model, h->pl.contact, h->persist_npb);
  if (!fn) return;
  h->persist_fwd_members = persist_members_that_fit(h, fn, h->persist_npb);
  if (!persist_members_ok(h, h->persist_fwd_members)) return;
  if (h->d_ring.ensure((size_t)kPRing * h->pl.batch * h->pl.n_blocks * kPos) != hipSuccess) { (void)hipGetLastError(); return; }
  h->persist_fwd = true;
}
void persist_plan_adj(dfx_handle* h, const DevCtx& c) {
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
int* persist_give_up_word(dfx_handle* h) { return reinterpret_cast<int*>(h->flag_stage.p) + 2; }   // (word 0: non-finite flag, word 1: touched flag)
int ensure_flags(dfx_handle* h) {
  HIP_OK(h->flag_stage.ensure(64 + sizeof(int) * (size_t)h->pl.batch));
  return 0;
}
int* member_flags(dfx_handle* h) { return reinterpret_cast<int*>(h->flag_stage.p) + 16; }
#ifdef DFX_PERSIST_TIMING
unsigned* persist_dbg_buffer() { static unsigned* p = nullptr; if (!p) { (void)hipHostMalloc((void**)&p, 8 * 4 * 8192 * sizeof(unsigned), hipHostMallocDefault); memset(p, 0, 8 * 4 * 8192 * sizeof(unsigned)); } return p; }
#endif
int persist_spin_limit(const dfx_handle* h) { return h->spin_limit > 0 ? h->spin_limit : kSpinLimit; }
void persist_fell_back(dfx_handle* h) {
  h->persist_off = true;
  h->persist_fwd = h->persist_adj = false;
  static bool said = false;
  if (!said) fprintf(stderr, "[dfx] a persistent launch could not get all its workgroups resident (another process on the device?): this engine keeps one "
                             "launch per stage from now on; the solve is run again that way\n");
  said = true;
}
void persist_forget(dfx_handle* h) {
  std::lock_guard<std::mutex> lk(g_persist_mu);
  auto& fl = g_persist_inflight[h->device & 63];
  for (size_t i = 0; i < fl.size();) {
    bool mine = fl[i].st == h->stream;
    for (auto& g : h->groups) mine = mine || fl[i].st == g.stream || fl[i].st == g.stream2;
    if (mine) { (void)hipEventSynchronize(fl[i].ev); g_persist_events[h->device & 63].push_back(fl[i].ev); fl.erase(fl.begin() + i); } else ++i;
  }
}
// one launch, admitted by the slot account above
static void launch_persist(dfx_handle* h, const void* fn, hipStream_t st, void** args, int grid, int need) {
  std::lock_guard<std::mutex> lk(g_persist_mu);
  const int d = h->device & 63;
  auto& fl = g_persist_inflight[d];
  for (size_t i = 0; i < fl.size();)          // retire what has finished
    if (hipEventQuery(fl[i].ev) == hipSuccess) { g_persist_events[d].push_back(fl[i].ev); fl.erase(fl.begin() + i); } else { (void)hipGetLastError(); ++i; }
  for (;;) {
    // the other streams' largest outstanding needs (launches of one stream never overlap)
    std::map<hipStream_t, std::pair<int, size_t>> other;       // stream -> (largest need, index of its latest entry)
    for (size_t i = 0; i < fl.size(); ++i)
      if (fl[i].st != st) { auto& o = other[fl[i].st]; o.first = std::max(o.first, fl[i].slots); o.second = i; }
    int total = need;
    for (auto& kv : other) total += kv.second.first;
    if (total <= kPersistSlots || other.empty()) break;
    // over the budget: queue behind everything the stream whose latest launch is oldest has outstanding, and forget its entries
    auto victim = other.begin();
    for (auto it = other.begin(); it != other.end(); ++it) if (it->second.second < victim->second.second) victim = it;
    (void)hipStreamWaitEvent(st, fl[victim->second.second].ev, 0);
    const hipStream_t vs = victim->first;
    for (size_t i = 0; i < fl.size();)
      if (fl[i].st == vs) { fl[i].st = st; ++i; } else ++i;      // (ordered before us from now on: they count as this stream's)
  }
  (void)hipLaunchKernel(fn, dim3(grid), dim3(kPersistThreads), args, 0, st);
  hipEvent_t ev = nullptr;
  if (!g_persist_events[d].empty()) { ev = g_persist_events[d].back(); g_persist_events[d].pop_back(); }
  else (void)hipEventCreateWithFlags(&ev, hipEventDisableTiming);
  (void)hipEventRecord(ev, st);
  fl.push_back({ev, st, need});
  h->launches++;
}
// one segment of the group's members: the first ring places poisoned, then the whole segment in one launch per `per_launch` members
static void launch_segment_persist(dfx_handle* h, const DevCtx& c, hipStream_t st, int nm, int n_steps, bool reverse, double* ring = nullptr) {
  if (!ring) ring = h->d_ring.p;
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
  const int n_chunks = (nm + per - 1) / per, even = (nm + n_chunks - 1) / n_chunks;      // members that do not fit at once: equal launches
  for (int off = 0; off < nm; off += even) {
    const int cnt = std::min(even, nm - off);
    DevCtx cc = c;
    cc.m0 = c.m0 + off;
    int grid = 0, per_cu = 0;
    int xcd_wg = 0;
    persist_shape(h, npb, cnt, &grid, &per_cu, fn, &xcd_wg);
    dfx_persist::launch_ring_poison(st, ring, h->pl.batch, h->pl.n_blocks, cc.m0, cnt, kPos);
    h->launches++;
    PersistArgs pa;
    pa.ring = ring; pa.give_up = persist_give_up_word(h); pa.n_steps = n_steps; pa.nm = cnt; pa.waves_per_member = h->persist_wpm; pa.spin_limit = persist_spin_limit(h); pa.xcd_wg = xcd_wg;
#ifdef DFX_PERSIST_TIMING
    pa.dbg = ((reverse ? getenv("DFX_TIMING_REVERSE") != nullptr : getenv("DFX_TIMING_REVERSE") == nullptr) && grid <= 8192) ? persist_dbg_buffer() : nullptr;
#endif
    void* args_f[] = {&cc, &pcf, &pa};
    void* args_r[] = {&cc, &pca, &pa};
    launch_persist(h, fn, st, reverse ? args_r : args_f, grid, per_cu * persist_wg_slots(fn));
  }
}
static void launch_fwd_persist(dfx_handle* h, const DevCtx& c, hipStream_t st, int nm, int n_steps) { launch_segment_persist(h, c, st, nm, n_steps, false); }

// ---- the adaptive controller inside the stage loop (dfx_persist_dense.h) and the reverse sweep of the steps it keeps ------------------
// forward: quad mapping (the controller's reductions are wave-wide), one member group; members that do not fit the chip at once run in
// further launches (every launch carries its members through up to `max_attempts` attempts on their own clocks)
bool persist_adaptive_plan(dfx_handle* h) {
  h->persist_fwd = false;
  if (!persist_shape_ok(h) || h->groups.size() != 1 || h->pl.tab.s != 6) return false;
  const void* fn = dfx_persist::adaptive_fwd_kernel(h->pl.model, h->pl.contact);
  if (!fn) return false;
  h->persist_npb = 4;
  h->persist_wpm = (h->pl.n_slots + 63) / 64;
  h->persist_fwd_members = persist_members_that_fit(h, fn, 4);
  if (h->persist_fwd_members <= 0 || (h->pl.batch + h->persist_fwd_members - 1) / h->persist_fwd_members > 8) return false;
  const size_t B = h->pl.batch;
  if (h->d_ring.ensure((size_t)kPRing * B * h->pl.n_blocks * kPos) != hipSuccess || h->d_err3.ensure(3 * B * (size_t)h->persist_wpm) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  h->persist_fwd = true;
  return true;
}
void launch_adaptive_persist(dfx_handle* h, const DevCtx& c, hipStream_t st, int max_attempts, AdaptLoopArgs aa) {
  const void* fn = dfx_persist::adaptive_fwd_kernel(h->pl.model, h->pl.contact);
  const Dopri D = make_dopri();
  AdaptLoopCoef pc;
  memset(&pc, 0, sizeof(pc));
  for (int i = 0; i < 7; ++i) {
    pc.e[i] = D.e[i]; pc.ee[i] = D.ee[i]; pc.cm[i] = D.cm[i]; pc.cma[i] = D.cma[i]; pc.c[i] = D.c[i];
    for (int l = 0; l < 7; ++l) { pc.a[i][l] = D.a[i][l]; pc.aa[i][l] = D.aa[i][l]; }
  }
  const int nm = h->pl.batch, per = h->persist_fwd_members;
  const int n_chunks = (nm + per - 1) / per, even = (nm + n_chunks - 1) / n_chunks;
  aa.err = h->d_err3.p;
  (void)hipMemsetAsync(h->d_err3.p, 0xFF, sizeof(double) * 3 * (size_t)nm * h->persist_wpm, st);       // every partial: poison
  for (int off = 0; off < nm; off += even) {
    const int cnt = std::min(even, nm - off);
    DevCtx cc = c;
    cc.m0 = c.m0 + off;
    int grid = 0, per_cu = 0;
    int xcd_wg = 0;
    persist_shape(h, 4, cnt, &grid, &per_cu, fn, &xcd_wg);
    dfx_persist::launch_ring_poison(st, h->d_ring.p, h->pl.batch, h->pl.n_blocks, cc.m0, cnt, kPos);
    h->launches++;
    PersistArgs pa;
    pa.ring = h->d_ring.p; pa.give_up = persist_give_up_word(h); pa.n_steps = max_attempts; pa.nm = cnt; pa.waves_per_member = h->persist_wpm; pa.spin_limit = persist_spin_limit(h); pa.xcd_wg = xcd_wg;
#ifdef DFX_PERSIST_TIMING
    pa.dbg = nullptr;
#endif
    void* args[] = {&cc, &pc, &pa, &aa};
    launch_persist(h, fn, st, args, grid, per_cu * persist_wg_slots(fn));
  }
}
// reverse: the records build's conditions (persist_plan_adj), the DENSE kernel's registers
bool persist_plan_adj_dense(dfx_handle* h, const DevCtx& c) {
  h->persist_adj = false;
  if (!persist_common_ok(h, c)) return false;
  if (c.rps <= 1 || c.g_b || c.AD || !c.lam_pairs) return false;
  const void* fn = dfx_persist::adj_dense_kernel(h->pl.model, h->pl.contact, h->persist_npb);
  if (!fn) return false;
  h->persist_adj_members = persist_members_that_fit(h, fn, h->persist_npb);
  if (h->persist_adj_members <= 0 || (h->pl.batch + h->persist_adj_members - 1) / h->persist_adj_members > 8) return false;
  if (h->d_ring.ensure((size_t)kPRing * h->pl.batch * h->pl.n_blocks * kPos) != hipSuccess) { (void)hipGetLastError(); return false; }
  h->persist_adj = true;
  return true;
}
void launch_adj_dense_persist(dfx_handle* h, const DevCtx& c, hipStream_t st, int n_steps, DenseCtx dn) {
  const int npb = h->persist_npb, per = h->persist_adj_members, nm = h->pl.batch;
  const void* fn = dfx_persist::adj_dense_kernel(h->pl.model, h->pl.contact, npb);
  PersistAdjCoef pca;
  memset(&pca, 0, sizeof(pca));
  for (int i = 0; i < h->pl.tab.s && i < kPersistStages; ++i) {
    const AdjCoef ac = adj_coef(h->pl.tab, i);
    for (int jj = 0; jj <= kPersistStages; ++jj) { pca.col[i][jj] = ac.col[jj]; pca.cur[i][jj] = ac.cur[jj]; }
    pca.c[i] = ac.c_i;
  }
  const int n_chunks = (nm + per - 1) / per, even = (nm + n_chunks - 1) / n_chunks;
  for (int off = 0; off < nm; off += even) {
    const int cnt = std::min(even, nm - off);
    DevCtx cc = c;
    cc.m0 = c.m0 + off;
    int grid = 0, per_cu = 0;
    int xcd_wg = 0;
    persist_shape(h, npb, cnt, &grid, &per_cu, fn, &xcd_wg);
    dfx_persist::launch_ring_poison(st, h->d_ring.p, h->pl.batch, h->pl.n_blocks, cc.m0, cnt, kPos);
    h->launches++;
    PersistArgs pa;
    pa.ring = h->d_ring.p; pa.give_up = persist_give_up_word(h); pa.n_steps = n_steps; pa.nm = cnt; pa.waves_per_member = h->persist_wpm; pa.spin_limit = persist_spin_limit(h); pa.xcd_wg = xcd_wg;
#ifdef DFX_PERSIST_TIMING
    pa.dbg = nullptr;
#endif
    void* args[] = {&cc, &pca, &pa, &dn};
    launch_persist(h, fn, st, args, grid, per_cu * persist_wg_slots(fn));
  }
}
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
int step_units(const dfx_handle* h, int kind) { return (kind == 0 ? h->pair_fwd : h->pair_adj) ? h->pl.tab.s / 2 : h->pl.tab.s; }
static void launch_fwd_unit(dfx_handle* h, const DevCtx& c, hipStream_t st, dim3 grid, int u, int j) {
  if (!h->pair_fwd) { launch_fwd_step_stage(h, c, st, grid, u, j); return; }
#ifdef DFX_EXPERIMENTAL
  const int s = h->pl.tab.s, i = 2 * u, nm = (int)grid.y;
  // records checkpoint: every record lives in the trajectory; otherwise the records ping-pong between stage buffers 1 and 2 and the
  // step state between buffers 0 and 3 (k_fwd_pair resolves buffer 0 by the parity of the step)
  if (c.rps > 1) launch_fwd_pair(h, c, st, nm, i, j, -1 - i, -1 - (i + 1), -1 - (i + 2), -1, 0);
  else launch_fwd_pair(h, c, st, nm, i, j, u == 0 ? 0 : 1 + ((u - 1) & 1), -1, i + 2 == s ? 0 : 1 + (u & 1), 0, (i + 2 == s && c.traj) ? 1 : 0);
#endif
}
// reverse unit u counts from the END of the step (u = 0: the last stage / pair)
static void launch_adj_unit(dfx_handle* h, const DevCtx& c, hipStream_t st, dim3 grid, int u, int j) {
  const int s = h->pl.tab.s;
  if (!h->pair_adj) { const int i = s - 1 - u; launch_adj(h, c, st, grid, i, j, adj_in_buf(c, i), -1, 0); return; }
#ifdef DFX_EXPERIMENTAL
  launch_adj_pair(h, c, st, (int)grid.y, s - 1 - 2 * u, j);
#endif
}

// the time functions of the segment the group's cursor now points at, for every step and stage time (k_fn_table); DFX_FN_TABLE=0:
// the lanes of driven / loaded blocks evaluate them themselves, as in rounds 1-2
bool use_fn_table(const dfx_handle* h) {
  const char* e = getenv("DFX_FN_TABLE");
  return h->pl.n_fns > 0 && !h->adaptive && h->d_fn_tab.p && !(e && e[0] == '0');
}
void launch_fn_table(dfx_handle* h, const DevCtx& c, hipStream_t st, int nm, int n_steps) {
  if (!c.fn_tab) return;
  StageTimes tms;
  for (int r = 0; r < kFnRows; ++r) tms.c[r] = r <= h->pl.tab.s ? h->pl.tab.c[r] : 0.0;
  const int total = n_steps * (h->pl.tab.s + 1) * h->pl.n_fns;
  hipLaunchKernelGGL(k_fn_table, dim3((total + 63) / 64, nm), dim3(64), 0, st, c, tms, n_steps, const_cast<double*>(c.fn_tab));
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
bool solve_is_eager(const dfx_handle* h) {
  if (!h->use_graph) return true;
  if (const char* e = getenv("DFX_EAGER_STEPS")) return h->n_total <= atoll(e);
  const long long waves = (long long)h->pl.batch * ((h->pl.n_slots + 63) / 64);
  return waves >= 2048 || h->n_total <= 128;
}

void enqueue_interleaved(dfx_handle* h, const DevCtx& cbase, int n_steps, int kind, int seg_index) {
  const int ng = (int)h->groups.size();
  if (kind == 1 && !cbase.AD && cbase.rps == 1) {      // recompute chains (events per group): group by group
    for (int gi = 0; gi < ng; ++gi) enqueue_segment(h, cbase, gi, n_steps, kind);
    return;
  }
  std::vector<DevCtx> cg(ng, cbase);
  for (int gi = 0; gi < ng; ++gi) {
    cg[gi] = group_ctx(h, cbase, gi);
    if (seg_index >= 0) hipLaunchKernelGGL(k_set_seg, dim3(1), dim3(1), 0, h->groups[gi].stream, (const Seg*)h->d_segs.p, seg_index, h->d_cur.p + gi);
    else hipLaunchKernelGGL(k_tick, dim3(1), dim3(1), 0, h->groups[gi].stream, (const Seg*)h->d_segs.p, h->d_seg_idx.p + 2 + gi, kind == 0 ? 1 : -1,
        h->d_cur.p + gi);
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

// ---- segments level where the persistent loop serves both sweeps: the re-run of piece k-1 beside the reverse stages of piece k ---------
// A piece's records are rebuilt by one cheap forward launch per segment and then read by its reverse launches; neither fills the chip
// (one 128x128 design: one wave per SIMD each), so the re-run of the NEXT piece (the one before it in time) runs on a second stream into
// a second record buffer while this piece is reversed.  Both launches must be resident together: the slot account of launch_persist
// would otherwise queue one behind the other (correct, but then the second buffer buys nothing) -- asked here, before anything is
// allocated.  DFX_SEG_OVERLAP=0 switches it off (A/B runs, and the bit-identity test against the serial order).
bool seg_overlap_plan(dfx_handle* h, const DevCtx& c) {
  const char* e = getenv("DFX_SEG_OVERLAP");
  if ((e && e[0] == '0') || !h->segments || !h->persist_fwd || !h->persist_adj || h->groups.size() != 1 || h->pieces.size() < 2) return false;
  const int npb = h->persist_npb, nm = h->pl.batch;
  const void* ff = dfx_persist::fwd_kernel(h->pl.model, h->pl.contact, npb);
  const void* fr = dfx_persist::adj_kernel(h->pl.model, h->pl.contact, npb);
  if (h->persist_fwd_members < nm || h->persist_adj_members < nm) return false;       // one launch per segment each
  int grid = 0, per_cu_f = 0, per_cu_r = 0, xf = 0, xr = 0;
  persist_shape(h, npb, nm, &grid, &per_cu_f, ff, &xf);
  persist_shape(h, npb, nm, &grid, &per_cu_r, fr, &xr);
  if (per_cu_f * persist_wg_slots(ff) + per_cu_r * persist_wg_slots(fr) > kPersistSlots) return false;
  const int of = persist_wg_per_cu(ff), orv = persist_wg_per_cu(fr), per_cu = std::max(per_cu_f, per_cu_r);
  if ((of && 2 * per_cu > of) || (orv && 2 * per_cu > orv)) return false;       // (occupancy by LDS / scratch: room for both with a margin)
  size_t steps = 0;
  for (const auto& pc : h->pieces) { size_t n = 0; for (int si = pc.first; si <= pc.last; ++si) n += h->segs[si].n_steps; steps = std::max(steps, n); }
  const size_t rec = (size_t)h->pl.n_blocks * kStep;
  const size_t want = (size_t)nm * (steps * h->pl.tab.s + 1) * rec;
  if (h->d_traj2.n < want) {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return false; }
    if ((want - h->d_traj2.n) * sizeof(double) + total_b / 20 > free_b) return false;
  }
  if (h->d_traj2.ensure(want) != hipSuccess || h->d_ring2.ensure((size_t)kPRing * nm * h->pl.n_blocks * kPos) != hipSuccess ||
      (c.fn_tab && h->d_fn_tab2.ensure(h->d_fn_tab.n) != hipSuccess)) { (void)hipGetLastError(); return false; }
  for (int b = 0; b < 2; ++b) {
    if (!h->ev_rebuilt[b] && hipEventCreateWithFlags(&h->ev_rebuilt[b], hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (!h->ev_reversed[b] && hipEventCreateWithFlags(&h->ev_reversed[b], hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return false; }
  }
  return true;
}
// one segment of the re-run on the second stream: its own cursor, time-function table and ring (cf carries cursor, table and record base)
void enqueue_rerun_segment(dfx_handle* h, const DevCtx& cf, hipStream_t st, int seg_index) {
  const int n_steps = h->segs[seg_index].n_steps;
  hipLaunchKernelGGL(k_set_seg, dim3(1), dim3(1), 0, st, (const Seg*)h->d_segs.p, seg_index, const_cast<Seg*>(cf.cur));
  h->launches++;
  launch_fn_table(h, cf, st, h->pl.batch, n_steps);
  launch_segment_persist(h, cf, st, h->pl.batch, n_steps, false, h->d_ring2.p);
}

int run_segment(dfx_handle* h, const DevCtx& c, int gi, int n_steps, int kind) {
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
  // launches in the graph
  const long long per = 1 + (c.fn_tab ? 1 : 0) + (long long)n_steps * ((kind == 0 || c.AD || c.rps > 1) ? step_units(h, kind) : 2 * s - 1);
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
int fork_groups(dfx_handle* h) {
  HIP_OK(hipEventRecord(h->ev_fork, h->stream));
  for (auto& g : h->groups) if (g.stream != h->stream) HIP_OK(hipStreamWaitEvent(g.stream, h->ev_fork, 0));
  return 0;
}
// ... and the main stream continue after all of them
int join_groups(dfx_handle* h) {
  for (auto& g : h->groups) {
    if (g.stream == h->stream) continue;
    HIP_OK(hipEventRecord(g.done, g.stream));
    HIP_OK(hipStreamWaitEvent(h->stream, g.done, 0));
  }
  return 0;
}
