#define DFX_ABI_LAYOUT_IMPL      // include/dfx.h then carries the body of dfx_abi_layout
// engine_abi.hip -- libdfx host side: create / destroy / set_params / reserve / share_checkpoint, the test hooks (dfx_rhs, dfx_rhs_vjp, dfx_energy),
// post-processing, downloads
// (one of five translation units; shared declarations in dfx_engine.h, the design in DESIGN.md section 3)
#include "dfx_engine.h"
#include "dfx_hostpar.h"
#include "dfx_design.h"

using namespace dfx_persist;

static std::string g_create_error;

// design -> geometry and its cotangent, on the host (dfx_design.h)
int dfx_design_forward(const dfx_design_map* map, const double* design, int32_t batch, double density, double* block_centroids,
                       double* centroid_node_vectors, double* inertia, double* void_angle0) {
  return dfx_design::forward(map, design, batch, density, block_centroids, centroid_node_vectors, inertia, void_angle0);
}
int dfx_design_vjp(const dfx_design_map* map, const double* design, int32_t batch, double density, const double* centroid_node_vectors_bar,
                   const double* block_centroids_bar, const double* inertia_bar, const double* void_angle0_bar, double* design_bar) {
  return dfx_design::vjp(map, design, batch, density, centroid_node_vectors_bar, block_centroids_bar, inertia_bar, void_angle0_bar, design_bar);
}


int dfx_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

#ifdef DFX_EXPERIMENTAL
const char* dfx_version(void) { return "dfx-hip-gfx950 0.3.0+experimental"; }      // (pair launches, tile kernels, DFX_TEST_FREE_BYTES compiled in)
#else
const char* dfx_version(void) { return "dfx-hip-gfx950 0.3.0"; }
#endif

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
  // the persistent loop's hand-off assumes that the slot a lane watches watches it back (dfx_persist.h): true for every bond list with one
  // ligament per node by construction; checked, not assumed (round-5 advice) -- a plan that breaks it keeps one launch per stage
  for (int sl = 0; sl < pl.n_slots && !h->persist_off; ++sl) {
    const int info = pl.slot_info[sl];
    if (info >= 0 && (pl.slot_info[info >> 1] < 0 || (pl.slot_info[info >> 1] >> 1) != sl)) h->persist_off = true;
  }
  *out = h;
  return 0;
}

int dfx_member_status(dfx_handle* h, int32_t* status) {
  for (int m = 0; m < h->pl.batch; ++m) status[m] = m < (int)h->member_status.size() ? h->member_status[m] : 0;
  return 0;
}
int dfx_set_failure_policy(dfx_handle* h, int32_t isolate) { h->isolate_failures = isolate != 0; return 0; }
int dfx_test_set_spin_limit(dfx_handle* h, int32_t polls) { h->spin_limit = polls > 0 ? polls : 0; return 0; }

int dfx_destroy(dfx_handle* h) {
  if (!h) return 0;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  persist_forget(h);           // (round-5 advice: the account of persistent launches in flight must not keep streams of a handle that is gone)
  drop_graphs(h);
  h->d_out_ptr.release(); h->d_nacc.release(); h->d_theta.release(); h->d_dw.release(); h->d_err3.release();
  h->d_ovf_ptr.release(); h->d_ovf_info.release(); h->d_ovf_bond.release(); h->d_ovf_p.release(); h->d_ovf_g.release();
  h->d_lig_slots.release(); h->d_lig_tab.release(); h->d_lig_p.release(); h->d_lig_l.release(); h->d_lig_k.release(); h->d_lig_phi.release();
  h->d_lig_g.release(); h->d_lig_gphi.release();
  h->d_slot_info.release(); h->d_block_special.release(); h->d_special.release(); h->d_slot_bond.release(); h->d_touch.release(); h->zero_phi.release();
  h->d_out_r.release(); h->d_out_phi.release(); h->d_out_lam.release(); h->d_resp.release();
  h->d_p_r.release(); h->d_p_l.release(); h->d_p_k.release(); h->d_p_phi.release(); h->d_cst.release(); h->d_l_dict.release(); h->d_l_idx.release();
  h->d_inv_m.release(); h->d_damping.release(); h->d_fns.release(); h->d_p_c.release(); h->d_g_c.release();
  for (int f = 0; f < DFX_MAX_FNS; ++f) h->d_fn_table[f].release();
  h->d_segs.release(); h->d_cur.release(); h->d_seg_idx.release(); h->d_clock.release(); h->d_err_partial.release(); h->d_ts.release();
      h->d_step_counts.release(); h->d_acc_times.release(); h->d_tsteps.release();
  if (--h->ck->users == 0) { h->ck->traj.release(); h->ck->AD.release(); delete h->ck; }
  h->d_traj2.release(); h->d_ring2.release(); h->d_fn_tab2.release();
  for (int b = 0; b < 2; ++b) { if (h->ev_rebuilt[b]) (void)hipEventDestroy(h->ev_rebuilt[b]); if (h->ev_reversed[b]) (void)hipEventDestroy(h->ev_reversed[b]); }
  h->d_ring.release(); h->d_fn_tab.release(); h->d_POS.release(); h->d_VEL.release(); h->d_A.release(); h->d_state0.release(); h->d_fields.release();
  h->d_YB.release(); h->d_LAM.release(); h->d_W.release(); h->d_KQ.release(); h->d_G.release(); h->d_restart.release();
  h->d_g_r.release(); h->d_g_phi.release(); h->d_g_b.release(); h->d_blk_m.release(); h->d_blk_c.release(); h->d_fn_g.release();
  h->d_tmp.release(); h->d_obj.release(); h->d_target.release(); h->stage.release(); h->obj_stage.release(); h->flag_stage.release();
  for (auto& gr : h->groups) {
    for (auto e : gr.ev_a) (void)hipEventDestroy(e);
    for (auto e : gr.ev_b) (void)hipEventDestroy(e);
    if (gr.stream2) (void)hipStreamDestroy(gr.stream2);
    if (gr.done) (void)hipEventDestroy(gr.done);
    if (gr.stream && gr.stream != h->stream) (void)hipStreamDestroy(gr.stream);
  }
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
    // images -> pinned staging area on a handful of threads (one thread copies at ~10 GB/s: 8 ms of the 12 ms this call took for 32 x 128 x 128),
    // then one DMA per image
    struct Job { char* dst; const char* src; size_t bytes; };
    std::vector<Job> jobs;
    constexpr size_t kChunk = (size_t)1 << 20;
    auto stage_in = [&](size_t at, const void* from, size_t bytes) {
      for (size_t o = 0; o < bytes; o += kChunk) jobs.push_back({h->stage.p + at + o, static_cast<const char*>(from) + o, std::min(kChunk, bytes - o)});
    };
    size_t off = 0, offs[NA] = {0};
    if (pp.l_dict_ok) {
      HIP_OK(h->d_l_idx.ensure(pp.l_idx.size()));
      stage_in(0, pp.l_idx.data(), pp.l_idx.size());
      off = (pp.l_idx.size() + 63) & ~(size_t)63;
    }
    for (int i = 0; i < NA; ++i) {
      if (!src[i] || src[i]->empty()) continue;
      HIP_OK(dst[i]->ensure(src[i]->size()));
      offs[i] = off;
      stage_in(off, src[i]->data(), src[i]->size() * sizeof(double));
      off += src[i]->size() * sizeof(double);
    }
    dfx_hostpar::for_each((int)jobs.size(), kChunk, [&](int j) { memcpy(jobs[j].dst, jobs[j].src, jobs[j].bytes); });
    if (pp.l_dict_ok) HIP_OK(hipMemcpyAsync(h->d_l_idx.p, h->stage.p, pp.l_idx.size(), hipMemcpyHostToDevice, h->stream));
    for (int i = 0; i < NA; ++i) {
      if (!src[i] || src[i]->empty()) continue;
      HIP_OK(hipMemcpyAsync(dst[i]->p, h->stage.p + offs[i], src[i]->size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
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
  hipLaunchKernelGGL(k_response, grid, dim3(kThreads), 0, h->stream, c, (const double*)h->d_fields.p, (const int32_t*)h->d_slot_bond.p,
      (const int32_t*)h->d_ovf_bond.p, (int)nbd,
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
#define DFX_EN_CASE(M) case M: if (pl.contact == 2) hipLaunchKernelGGL((k_energy<M, 2>), slot_grid(h), dim3(kThreads), 0, h->stream, c, h->d_tmp.p); \
    else if (pl.contact) hipLaunchKernelGGL((k_energy<M, 1>), slot_grid(h), dim3(kThreads), 0, h->stream, c, h->d_tmp.p); \
    else hipLaunchKernelGGL((k_energy<M, 0>), slot_grid(h), dim3(kThreads), 0, h->stream, c, h->d_tmp.p); break;
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
