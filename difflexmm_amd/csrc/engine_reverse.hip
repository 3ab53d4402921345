// engine_reverse.hip -- libdfx host side: the reverse sweep (dfx_adjoint, the kinetic-energy objective calls, the fused forward + reverse call), gradient
// collection
// (one of five translation units; shared declarations in dfx_engine.h, the design in DESIGN.md section 3)
#include "dfx_engine.h"

using namespace dfx_persist;

int ensure_adjoint_buffers(dfx_handle* h) {
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
int zero_grad_accumulators(dfx_handle* h, double* extra, size_t n_extra, int cursor_value,
                                  const int32_t* targets, int n_target) {
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
#ifdef DFX_EXPERIMENTAL
  if (h->lig_ok && h->lig.g) { zero(h->d_lig_g.p, B * nb * 8); zero(h->d_lig_gphi.p, B * nb * 4); }
#endif
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
int collect_grads(dfx_handle* h, const dfx_grads* want, dfx_grads* grads, dfx_grads* views, bool with_state0) {
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
#ifdef DFX_EXPERIMENTAL
  if (h->lig_used) {        // the tile kernels accumulated ligament-major: fold into the slot-major accumulators read below
    DevCtx c = make_ctx(h);
    hipLaunchKernelGGL(k_lig_unpack, dim3((unsigned)((nb * 2 + kThreads - 1) / kThreads), (unsigned)B), dim3(kThreads), 0, h->stream, c,
                       (const int32_t*)h->d_lig_slots.p, (const double*)h->d_lig_g.p, pl.contact == DFX_CONTACT_ANGLE ? (const double*)h->d_lig_gphi.p
                           : (const double*)nullptr);
    h->lig_used = false;
  }
#endif
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
    if (ensure_flags(h)) return 2;
    // (word 0: the forward pass's flag)
    HIP_OK(hipMemcpyAsync(reinterpret_cast<int32_t*>(h->flag_stage.p) + 1, h->d_touch.p, sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
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

void set_grad_wishes(dfx_handle* h, const dfx_grads* g) {
  h->want_bond_grads = !g || g->reference_vector || g->k_bond || g->contact;
  h->want_fn_grads = !g || g->fn_params;
  h->want_damping_grads = !g || g->damping;
}


// reverse sweep with the output cotangents already in h->d_G
// accumulators_cleared: the caller's prelude launch has already zeroed the gradient accumulators and set the cursors (adjoint_kinetic
// does it in the launch that clears its cotangents) -- an argument, not handle state: a flag left behind by a call that failed half way
// made the next sweep skip its zeroing (round-4 advice)
static int run_adjoint(dfx_handle* h, const dfx_grads* want, dfx_grads* grads, dfx_grads* views, dfx_stats* stats, bool kinetic, int n_target,
                       bool accumulators_cleared = false) {
  if (h->adaptive_records) {       // the accepted steps of an adaptive solve: engine_dense.hip (accumulators cleared by the caller's prelude
    set_grad_wishes(h, want);      // launch, or here)
    if (!accumulators_cleared && zero_grad_accumulators(h, nullptr, 0, -1)) return 2;
    return run_adjoint_dense(h, want, grads, views, stats, kinetic, n_target);
  }
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
  bool overlapped = false;
  if (h->segments) {
    // output intervals backwards: records of interval k rebuilt by re-running its forward pass from the resident output row k
    // (bit-identical to the first pass: same state, same arithmetic), then its reverse stages read them
    const size_t nb6 = (size_t)pl.n_blocks * 6;
    // (re)start every member group at the first step of segment si from a state row: stage buffer 0 and record 0 of that step
    auto restart = [&](const DevCtx& cc, const double* rows, long long row_stride, int si) {
      const Seg& s0 = h->segs[si];
      for (int gi = 0; gi < (int)h->groups.size(); ++gi) {
        const Group& gr = h->groups[gi];
        const DevCtx cg = group_ctx(h, cc, gi);
        hipLaunchKernelGGL(k_init, slot_grid(h, gr), dim3(kThreads), 0, gr.stream, cg, rows, s0.t_interval + s0.j0 * s0.h, 0, row_stride,
                           (long long)s0.base_step);
        hipLaunchKernelGGL(k_checkpoint0, dim3((unsigned)(((size_t)pl.n_blocks * kStep + kThreads - 1) / kThreads), (unsigned)gr.nm), dim3(kThreads), 0,
                           gr.stream, cg, (long long)s0.base_step);
        h->launches += 2;
      }
    };
    // the pieces backwards (one per output interval unless seg_chunk cuts it: engine_forward.hip).  The buffer holds the records of ONE piece:
    // its base is shifted so that the kernels keep indexing by the global step (a per-interval offset inside the kernels cost the forward
    // kernel two hot-path spills: profiles/r02_fwd_spill_regression.txt).  The records of a piece are rebuilt by re-running its forward pass
    // from the resident output row of its interval, or from the restart row the forward pass left (bit-identical to the first pass: a step
    // starts from nothing but its state, same arithmetic), then its reverse stages read them.
    const int np = (int)h->pieces.size();
    auto shifted = [&](double* buf, const dfx_handle::Piece& pc) {
      return buf - (size_t)h->segs[pc.first].base_step * (size_t)c.rps * pl.batch * ((size_t)pl.n_blocks * kStep);
    };
    if (seg_overlap_plan(h, c)) {
      // Persistent loop, launches that leave room for each other: piece k-1 is re-run on the group's second stream into the OTHER record
      // buffer while piece k is reversed on the first (engine_launch.hip, seg_overlap_plan).  Per buffer two events: records rebuilt ->
      // its reverse launches may start; reverse stages done -> the re-run after next may overwrite it.  The re-run has its own cursor,
      // time-function table and ring; the stage buffers belong to it alone (the reverse launches of the records level read records).
      Group& g0 = h->groups[0];
      const hipStream_t sf = g0.stream2, sr = g0.stream;
      double* bufs[2] = {h->ck->traj.p, h->d_traj2.p};
      DevCtx cf = group_ctx(h, c, 0);
      cf.cur = h->d_cur.p + 1;
      if (c.fn_tab) cf.fn_tab = h->d_fn_tab2.p;
      HIP_OK(hipEventRecord(h->ev_fork2, sr));
      HIP_OK(hipStreamWaitEvent(sf, h->ev_fork2, 0));
      auto rerun = [&](int pi) {
        const dfx_handle::Piece& pc = h->pieces[pi];
        const int b = (np - 1 - pi) & 1;
        if (np - 1 - pi >= 2) (void)hipStreamWaitEvent(sf, h->ev_reversed[b], 0);
        cf.traj = shifted(bufs[b], pc);
        const Seg& s0 = h->segs[pc.first];
        const double* rows = pc.row < 0 ? h->d_fields.p + (size_t)pc.interval * nb6 : h->d_restart.p + (size_t)pc.row * nb6;
        const long long stride = pc.row < 0 ? (long long)((size_t)Tn * nb6) : (long long)((size_t)h->n_restart_rows * nb6);
        hipLaunchKernelGGL(k_init, slot_grid(h, g0), dim3(kThreads), 0, sf, cf, rows, s0.t_interval + s0.j0 * s0.h, 0, stride, (long long)s0.base_step);
        hipLaunchKernelGGL(k_checkpoint0, dim3((unsigned)(((size_t)pl.n_blocks * kStep + kThreads - 1) / kThreads), (unsigned)g0.nm), dim3(kThreads), 0,
                           sf, cf, (long long)s0.base_step);
        h->launches += 2;
        for (int si = pc.first; si <= pc.last; ++si) enqueue_rerun_segment(h, cf, sf, si);
        (void)hipEventRecord(h->ev_rebuilt[b], sf);
      };
      rerun(np - 1);
      for (int pi = np - 1; pi >= 0; --pi) {
        if (pi > 0) rerun(pi - 1);
        const dfx_handle::Piece& pc = h->pieces[pi];
        const int b = (np - 1 - pi) & 1;
        (void)hipStreamWaitEvent(sr, h->ev_rebuilt[b], 0);
        c.traj = shifted(bufs[b], pc);
        for (int si = pc.last; si >= pc.first; --si) enqueue_interleaved(h, c, h->segs[si].n_steps, 1, si);
        (void)hipEventRecord(h->ev_reversed[b], sr);
      }
      overlapped = true;
    } else
    for (int pi = np - 1; pi >= 0; --pi) {
      const dfx_handle::Piece& pc = h->pieces[pi];
      c.traj = shifted(h->ck->traj.p, pc);
      if (pc.row < 0) restart(c, h->d_fields.p + (size_t)pc.interval * nb6, (long long)((size_t)Tn * nb6), pc.first);
      else restart(c, h->d_restart.p + (size_t)pc.row * nb6, (long long)((size_t)h->n_restart_rows * nb6), pc.first);
      for (int si = pc.first; si <= pc.last; ++si) enqueue_interleaved(h, c, h->segs[si].n_steps, 0, si);
      for (int si = pc.last; si >= pc.first; --si) enqueue_interleaved(h, c, h->segs[si].n_steps, 1, si);
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
#ifdef DFX_PERSIST_TIMING
  if (getenv("DFX_TIMING_REVERSE") && getenv("DFX_TIMING_WAVES")) {
    extern unsigned* persist_dbg_buffer();
    const unsigned* d = persist_dbg_buffer();
    const int nw = std::min(4096, (int)h->pl.batch * ((h->pl.n_slots + 63) / 64));
    float ms1 = 0.f; (void)hipEventElapsedTime(&ms1, h->ev2, h->ev3);
    fprintf(stderr, "[dfx] reverse loop per wave, last launch: before the poll / poll / Hessian-vector product / epilogue + rest, in counter ticks per stage (sweep %.3f us per stage):", 1e3 * ms1 / std::max<long long>(1, h->n_total * pl.tab.s));
    for (int w = 0; w < nw; ++w) { const unsigned* q = d + (size_t)w * 8; if (!q[7]) continue;
      fprintf(stderr, "%s%d:%u/%u/%u/%u", w % 6 ? "  " : "\n   ", w, q[0] / q[7], q[1] / q[7], q[2] / q[7], (q[3] + q[4] + q[5]) / q[7]); }
    fprintf(stderr, "\n");
  }
#endif
  if (*persist_give_up_word(h)) {
    // a wave of a persistent launch gave up.  A give-up inside the forward pass of the fused call (its flag is read only now) invalidates the
    // trajectory: the fused call runs everything again (-7).  Otherwise the records are good: the sweep alone is run again on stage launches
    if (h->defer_forward_sync) return -7;
    persist_fell_back(h);
    *persist_give_up_word(h) = 0;
    if (zero_grad_accumulators(h, nullptr, 0, nseg)) return 2;
    return run_adjoint(h, want, grads, views, stats, kinetic, n_target, true);
  }
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
    stats->streams = overlapped ? 2 : (int64_t)h->groups.size();     // (segments level, re-run beside the reverse stages: two streams)
    stats->stage_kernel_us = h->n_total ? 1e3 * ms / (double)(h->n_total * pl.tab.s * 2) : 0.0;
    stats->stage_checkpoint = c.AD ? 1 : 0;
    stats->checkpoint_records = h->segments ? 2 : (c.rps > 1 ? 1 : 0);
    stats->tile_kernels = h->persist_adj ? 3 : ((c.g_b || c.AD) ? 0 : kernel_build_code(h, c, h->lig_adj_used));
  }
  return 0;
}


static const char* kStaleCheckpoint =
    "the shared trajectory checkpoint was overwritten by a solve of another handle (dfx_share_checkpoint): run this handle's forward again";

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
  for (int pass = 0; pass < 2; ++pass) {
    h->defer_forward_sync = true;
    int rc = forward_grid_impl(h, state0, timepoints, n_timepoints, steps_per_interval, nullptr, 1, nullptr, nullptr, false);
    if (rc) { h->defer_forward_sync = false; return rc; }
    h->device_views = device_views != 0;
    rc = adjoint_kinetic(h, target_blocks, n_target, objective, want, nullptr, views, adjoint_stats);
    h->device_views = false;
    h->defer_forward_sync = false;
    HIP_OK(hipStreamSynchronize(h->stream));          // (already idle when the sweep returned normally)
    const int rcf = finish_forward(h, forward_stats);
    if ((rc == -7 || rcf == -7) && pass == 0) { persist_fell_back(h); continue; }      // not fully resident: once more, one launch per stage
    if (rc == -7 || rcf == -7) { h->err = kPersistGaveUp; return 2; }
    return rcf ? rcf : rc;
  }
  return 2;
}
