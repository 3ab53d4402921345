// variants/experimental/dfx_experimental_host.h -- host side of the two measured-slower experiments that stay in the tree behind -DDFX_EXPERIMENTAL (make experimental):
// two Runge-Kutta stages per launch on lattice windows (dfx_pair.h, DFX_PAIR=1) and every ligament evaluated once on lattice tiles
// (dfx_tile.h, DFX_TILE=1).  Included by engine_launch.hip only; the default libdfx.so contains none of it.
#pragma once

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
void setup_tiling(dfx_handle* h) {
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
void pair_plan(dfx_handle* h, const DevCtx& c) {
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
void setup_lig(dfx_handle* h) {
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
int lig_pack(dfx_handle* h) {
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
bool lig_fwd_ok(const dfx_handle* h, const DevCtx& c, int mode) {
  return h->lig_ok && h->lig.tab && !c.clock && !(mode & 2) && (c.fn_tab || h->pl.n_fns == 0);
}
bool lig_adj_ok(const dfx_handle* h, const DevCtx& c, int wbuf, int local_only) {
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

