// engine_forward.hip -- libdfx host side: the fixed-grid forward solve (dfx_forward, dfx_forward_grid, dfx_forward_grid_members): checkpoint levels,
// segments, snapshots
// (one of five translation units; shared declarations in dfx_engine.h, the design in DESIGN.md section 3)
#include "dfx_engine.h"

using namespace dfx_persist;

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

int choose_checkpoint(dfx_handle* h, long long n_steps, long long max_interval_steps) {
  const Plan& pl = h->pl;
  size_t dfx_test_free_bytes = 0;
#ifdef DFX_EXPERIMENTAL      // test hook (pretend that only so much HBM is free): experimental builds only
  { const char* t = getenv("DFX_TEST_FREE_BYTES"); dfx_test_free_bytes = t ? (size_t)atoll(t) : 0; }
#endif
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
  size_t want_seg = B * ((size_t)std::max<long long>(max_interval_steps, 1) * pl.tab.s + 1) * rec;
  h->seg_chunk = 0;
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
    // Where the persistent stage loop serves the solve (dfx_persist.h) and the records do not fit, the segments level comes next: its
    // forward re-run is one cheap persistent launch per segment and its reverse sweep reads records, i.e. stays persistent too -- the
    // stages / state levels would put the reverse sweep back on one launch per stage (one 128x128 design over the whole 50 000 steps:
    // 7.7 us per stage pair against 9.4)
    if (fits(grow(want_rec, have_t))) mode = kCkRecords;
    else if (persist_would_serve(h) && fits(grow(want_seg, have_t))) mode = kCkSegments;
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
    // An output interval whose stage records do not fit (few output times, many steps between them, many members) is re-run in pieces of
    // whole graph segments (kMaxGraphSteps steps): this forward pass leaves a restart state at every piece boundary (below), the reverse
    // sweep re-runs and reverses piece by piece (run_adjoint).  The piece: as long as fits next to what is allocated, at least one segment.
    // DFX_SEG_CHUNK_STEPS=n forces pieces of n steps (tests).
    long long piece = 0;
    if (const char* e = getenv("DFX_SEG_CHUNK_STEPS")) piece = std::max<long long>(kMaxGraphSteps, (atoll(e) / kMaxGraphSteps) * kMaxGraphSteps);
    else if (!fits(grow(want_seg, have_t)) && max_interval_steps > kMaxGraphSteps) {
      (void)fits(1);       // (queries the free memory if nothing has yet)
      const size_t free_now = dfx_test_free_bytes ? std::min<size_t>(free_b, dfx_test_free_bytes) : free_b;
      const size_t budget = free_now > total_b / 20 ? (free_now - total_b / 20) / sizeof(double) + have_t : have_t;
      const long long can = (long long)(budget / (B * rec) > 1 ? (budget / (B * rec) - 1) / pl.tab.s : 0);
      piece = std::max<long long>(kMaxGraphSteps, (can / kMaxGraphSteps) * kMaxGraphSteps);
    }
    if (piece > 0 && piece < max_interval_steps) {
      h->seg_chunk = piece;
      want_seg = B * ((size_t)piece * pl.tab.s + 1) * rec;
    }
    if (h->ck->traj.ensure(want_seg) != hipSuccess) {
      (void)hipGetLastError();
      h->err = "forward: cannot allocate the trajectory checkpoint: even its smallest form, the stage records of " + std::to_string(kMaxGraphSteps)
               + " steps of all members (" + std::to_string((want_seg * sizeof(double)) >> 20) + " MiB), does not fit the device next to what is "
               "allocated -- integrate fewer members per engine call (the problem layer runs a longer list of designs in calls of `batch`)";
      return done(-1);
    }
    return done(mode);
  }
  // allocate; a failed allocation falls back one level (forced modes included: the solve still runs)
  if (mode == kCkRecords && h->ck->traj.ensure(want_rec) != hipSuccess) { (void)hipGetLastError(); mode = kCkStages; }
  if (mode != kCkRecords && h->ck->traj.ensure(want_state) != hipSuccess) {
    (void)hipGetLastError();
    // (the free-memory query said it would fit: another process, or another thread of this one, took the memory in between)
    h->err = "forward: cannot allocate the trajectory checkpoint (the step states of all members: " + std::to_string((want_state * sizeof(double)) >> 20)
             + " MiB) although the device reported enough free memory a moment ago -- is another process allocating on this GPU?";
    return done(-1);
  }
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

int ensure_work_buffers(dfx_handle* h) {
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


// after the stream has been waited for: the non-finite flags of the forward pass (pinned words of flag_stage) and its statistics.
// Returns -7 when a wave of a persistent launch gave up: the caller latches the handle onto stage launches and runs the solve again.
int finish_forward(dfx_handle* h, dfx_stats* stats) {
#ifdef DFX_PERSIST_TIMING
  { extern unsigned* persist_dbg_buffer();
    const unsigned* d = persist_dbg_buffer();
    const int nw = std::min(4096, (int)h->pl.batch * ((h->pl.n_slots + 63) / 64));
    if (getenv("DFX_TIMING_WAVES") && d[7] > 0) {
      const int* w0 = persist_give_up_word(h);
      const double tk = 10.0 * w0[8] / std::max(1, w0[7]);
      fprintf(stderr, "[dfx] per wave: poll / ligament+contact / reduce+epilogue / rest ns per stage:");
      for (int w = 0; w < nw; ++w) { const unsigned* q = d + (size_t)w * 8; if (!q[7]) continue;
        fprintf(stderr, "%s%d:%.0f/%.0f/%.0f/%.0f", w % 6 ? "  " : "\n   ", w, tk * q[1] / q[7], tk * q[2] / q[7], tk * q[3] / q[7], tk * (q[0] + q[4] + q[5]) / q[7]); }
      fprintf(stderr, "\n");
    } }
  { const int* w = persist_give_up_word(h);
    if (w[9] > 0) { const double tk = 10.0 * w[8] / std::max(1, w[7]);      // ns per tick
      fprintf(stderr, "[dfx] forward loop, wave 0, last launch (%d stages, %.3f us per stage, %.2f ns per tick): pre-poll %.0f  poll %.0f  ligament %.0f  reduce+epilogue %.0f  sincos+ring store %.0f  checkpoint stores %.0f ns per stage\n",
              w[9], 1e-3 * 10.0 * w[8] / w[9], tk, tk * (unsigned)w[1] / w[9], tk * (unsigned)w[2] / w[9], tk * (unsigned)w[3] / w[9], tk * (unsigned)w[4] / w[9], tk * (unsigned)w[5] / w[9], tk * (unsigned)w[6] / w[9]); } }
#endif
  if (*persist_give_up_word(h)) { h->have_traj = false; return -7; }
  const int bad = *reinterpret_cast<const int*>(h->flag_stage.p);
  h->member_status.assign(h->pl.batch, 0);
  if (bad) {
    int first = -1, n_bad = 0;
    for (int m = 0; m < h->pl.batch; ++m)
      if (member_flags(h)[m]) { h->member_status[m] = 1; ++n_bad; if (first < 0) first = m; }
    if (!h->isolate_failures) {
      h->have_traj = false;
      h->err = "forward: non-finite state at output " + std::to_string(bad - 1) + " (unstable step size or contact blow-up) of " + std::to_string(n_bad) +
               " member(s), first: member " + std::to_string(std::max(first, 0)) + " -- dfx_member_status says which; dfx_set_failure_policy(h, 1) flags "
               "them instead of failing the call";
      return 3;
    }
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

// a non-finite initial state would travel through the persistent loop's hand-off ring as "record not there yet" (its poison is a NaN
// pattern) and is never a valid input: refused up front (round-5 advice)
static bool state_is_finite(const double* p, size_t n) {
  for (size_t i = 0; i < n; ++i) if (!std::isfinite(p[i])) return false;
  return true;
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
int forward_grid_impl(dfx_handle* h, const double* state0, const double* timepoints, int32_t n_timepoints,
                             const int32_t* steps_per_interval, const double* step_times, int32_t keep_trajectory,
                             double* fields, dfx_stats* stats, bool per_member) {
  HIP_OK(hipSetDevice(h->device));
  if (!h->have_params) { h->err = "forward: set_params first"; return 1; }
  h->adaptive = false;
  h->adaptive_records = false;
  if (n_timepoints < 1) { h->err = "forward: need >= 1 timepoint and >= 1 step per interval"; return 1; }
  if (state0 && !state_is_finite(state0, (size_t)h->pl.batch * h->pl.n_blocks * 6)) { h->err = "forward: state0 holds a non-finite value"; return 1; }
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
        if (tg[h->step0[k]] != timepoints[g * (size_t)Tn + k]) { h->err = "forward: step_times must contain every timepoint at the start of its interval";
            return 1; }
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
  // pieces of the segments level (one per interval unless seg_chunk cuts it) and their restart rows
  h->pieces.clear();
  h->n_restart_rows = 0;
  if (h->have_traj && h->segments)
    for (int k = 0; k + 1 < Tn; ++k)
      for (int si = h->seg_first[k]; si <= h->seg_last[k];) {
        int last = si;
        long long n = h->segs[si].n_steps;
        while (last + 1 <= h->seg_last[k] && (h->seg_chunk == 0 || n + h->segs[last + 1].n_steps <= h->seg_chunk)) n += h->segs[++last].n_steps;
        h->pieces.push_back({si, last, k, si == h->seg_first[k] ? -1 : h->n_restart_rows++});
        si = last + 1;
      }
  if (h->n_restart_rows) HIP_OK(h->d_restart.ensure(B * (size_t)h->n_restart_rows * nb * 6));
  HIP_OK(h->d_segs.ensure(std::max<size_t>(1, h->segs.size())));
  if (!h->segs.empty())
    HIP_OK(hipMemcpyAsync(h->d_segs.p, h->segs.data(), sizeof(Seg) * h->segs.size(), hipMemcpyHostToDevice, h->stream));
  // [0] unused, [1] non-finite flag (adaptive solves; fixed grids: pinned, below), [2+g] segment cursor of group g
  std::vector<int> cursors(2 + kMaxGroups, -1);
  cursors[1] = 0;
  // the non-finite flag of a fixed-grid solve lives in pinned host memory: k_snapshot stores into it directly (rare, any writer wins)
  // and the host reads it after its wait -- no device-to-host copy on the stream between the forward pass and whatever follows it
  if (ensure_flags(h)) return 2;
  int* const bad_flag = reinterpret_cast<int*>(h->flag_stage.p);
  int* const bad_m = member_flags(h);
  *bad_flag = 0;
  memset(bad_m, 0, sizeof(int) * B);
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
  hipLaunchKernelGGL(k_snapshot, g3, dim3(kThreads), 0, h->stream, c, h->d_fields.p, 0, bad_flag, 0, 0LL, bad_m);
  HIP_OK(hipEventRecord(h->ev0, h->stream));
  if (fork_groups(h)) return 2;
  const bool eager = solve_is_eager(h) || h->segments || h->persist_fwd;      // (a persistent segment is three launches: nothing to replay)
  size_t next_piece = 0;
  for (size_t si = 0; si < h->segs.size(); ++si) {
    const Seg& sg = h->segs[si];
    while (next_piece < h->pieces.size() && h->pieces[next_piece].first <= (int)si) ++next_piece;      // the first piece that starts after this segment
    if (eager) enqueue_interleaved(h, c, sg.n_steps, 0);
    for (int gi = 0; gi < (int)h->groups.size(); ++gi) {
      if (!eager) if (int rc = run_segment(h, c, gi, sg.n_steps, 0)) return rc;
      if (next_piece < h->pieces.size() && h->pieces[next_piece].row >= 0 && h->pieces[next_piece].first == (int)si + 1) {
        // segments level, an interval in pieces: the state the NEXT piece starts from, for the reverse sweep's re-run of that piece
        const Group& gr = h->groups[gi];
        DevCtx cs = group_ctx(h, c, gi);
        cs.n_timepoints = h->n_restart_rows;           // k_snapshot's row stride
        hipLaunchKernelGGL(k_snapshot, dim3(g3.x, gr.nm), dim3(kThreads), 0, gr.stream, cs, h->d_restart.p, h->pieces[next_piece].row, bad_flag,
                           pair_state_buf(h, h->segs[si + 1].base_step), (long long)h->segs[si + 1].base_step, bad_m);
        h->launches++;
      }
      if (sg.j0 + sg.n_steps == h->spis[sg.interval]) {   // buffer 0 holds the state at the end of the interval
        const Group& gr = h->groups[gi];
        // end of the interval: the state is in buffer 0, or (records checkpoint) only in the trajectory
        hipLaunchKernelGGL(k_snapshot, dim3(g3.x, gr.nm), dim3(kThreads), 0, gr.stream, group_ctx(h, c, gi), h->d_fields.p, sg.interval + 1,
                           bad_flag, c.rps > 1 ? -1 : pair_state_buf(h, h->step0[sg.interval + 1]),
                           (long long)h->step0[sg.interval + 1], bad_m);
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
  const int rcf = finish_forward(h, stats);
  if (rcf == -7) {      // a persistent launch was not fully resident: one launch per stage from now on, and this solve once more -- same process
    persist_fell_back(h);
    return forward_grid_impl(h, state0, timepoints, n_timepoints, steps_per_interval, step_times, keep_trajectory, fields, stats, per_member);
  }
  return rcf;
}
