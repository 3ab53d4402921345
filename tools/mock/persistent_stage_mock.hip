// Mock for the round-5 question: can ONE launch run many Runge-Kutta stages of a lattice that does not fill the chip, with waves
// handing their stage records to their <= 4 neighbour waves, faster than one launch per stage (7.0-7.7 us per stage for one 128x128
// system today)?  NOT the product: synthetic parameters and synthetic arithmetic with the product's shape --
//   lane = (block, node slot), 16 blocks per wave64, a wave = 16 consecutive blocks of one lattice row;
//   per stage and lane: the partner block's 32-byte record (two 16-byte loads), ~350 fp64 instructions (8 chains), three quad sums on DPP,
//   a DOF epilogue with the Runge-Kutta sums over the earlier stage accelerations, one sincos, and the new record: two 16-byte stores
//   per block (lanes 0 and 1) + one 8-byte velocity store per DOF lane -- all into a checkpoint that keeps EVERY stage record.
// Persistent form: a wave owns its 16 blocks for the whole launch; parameters, step state and stage accelerations stay in registers;
// the record of (step, stage) is stored ONCE, write-through (sc1), into its own place in the checkpoint, which the host has filled
// with a poison pattern (all-ones = a NaN no arithmetic produces); a lane reads its partner's record with sc1 loads until none of
// the four doubles is poison -- the data is the flag (an aligned 8-byte word is written by one store and is never seen torn:
// MI355X_MICROARCH.md, R2), no address is written twice per launch, so no stale copy can be mistaken for data.  Every spin is bounded.
// Baseline form: the same arithmetic, one launch per stage, plain loads and stores.  The two must agree in every word.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o persistent_stage_mock persistent_stage_mock.hip     (no contraction: the two
//   forms are different kernels, and only without it do they round alike -- any differing word is then a hand-off error)
//   ./persistent_stage_mock [members=1] [steps=200] [n1=128] [skew=0] [iters=22: 16 fp64 operations each] [nosleep=0] [selfpoison=0] [ring=0: places in the hand-off ring (needs selfpoison=1, >= 5)]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int kStages = 6;
constexpr int kAhead = 4;    // self-poison distance (stages)
typedef unsigned u32;
typedef unsigned long long u64;
typedef unsigned v4u __attribute__((ext_vector_type(4)));

struct Ctx {
  int n1, nb, members, steps, iters, skew, nosleep, selfpoison, ring;
  double* ringbuf;    // ring > 0: [ring][member][block][4] -- the records are handed over through a ring of `ring` places (self-poisoned); rec gets a plain copy
  double* rec;        // [ordinal = step * 6 + stage][member][block][4]   (ordinal steps*6 = the final state)
  double* vel;        // [ordinal][member][block][3]
  const double* par;  // [member][block*4 + slot][4]  synthetic per-slot parameters
  const double* dofp; // [member][block][3][2]  1/m, damping
  int* err;           // err[0] != 0: a spin gave up (value = 1 + ordinal)
};

__device__ __forceinline__ int neighbour(int b, int k, int n1) {
  int r = b / n1, c = b % n1;
  r += (k == 1) - (k == 3); c += (k == 0) - (k == 2);
  r = min(max(r, 0), n1 - 1); c = min(max(c, 0), n1 - 1);
  return r * n1 + c;
}
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double quad_sum(double v) { v += dpp_mov<0xB1>(v); v += dpp_mov<0x4E>(v); return v; }
template <int J> __device__ __forceinline__ double quad_bcast(double v) { return dpp_mov<J | (J << 2) | (J << 4) | (J << 6)>(v); }

__constant__ double c_cv[kStages][kStages], c_cq[kStages][kStages], c_c[kStages + 1];

// the synthetic "ligament": a function of the own record, the partner's record and the slot's parameters; ~8 * iters fp64 FMAs
__device__ __forceinline__ void ligament(const double o[4], const double p[4], const double par[4], int iters, double& fx, double& fy, double& fth) {
  double x[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) x[c] = 1e-3 * (p[c & 3] - o[c & 3]) + par[(c + 1) & 3] * 1e-3;
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int c = 0; c < 8; ++c) x[c] = fma(x[c], 0.96875, x[(c + 1) & 7] * 0.03125);
  }
  fx = x[0] + x[3] - x[6]; fy = x[1] - x[4] + x[7]; fth = x[2] + x[5];
}

struct LaneState { double q, v, invm, damp, a[kStages]; };

// DOF epilogue shared by both forms: stage acceleration, Runge-Kutta sums, next record value of this lane's DOF
__device__ __forceinline__ void epilogue(LaneState& S, int i, double dE, double v_i, double h, double& qnext, double& vnext) {
  const double a = (-dE - S.damp * v_i) * S.invm;
#pragma unroll
  for (int l = 0; l < kStages; ++l) S.a[l] = l == i ? a : S.a[l];
  double sv = 0.0, sq = 0.0;
#pragma unroll
  for (int l = 0; l < kStages; ++l) { sv += c_cv[i][l] * S.a[l]; sq += c_cq[i][l] * S.a[l]; }
  qnext = S.q + h * (c_c[i + 1] * S.v + h * sq);
  vnext = S.v + h * sv;
}

__device__ __forceinline__ double half_sin(double th) { const double x = 0.5 * th, x2 = x * x; return x * (1.0 + x2 * (-1.0 / 6 + x2 * (1.0 / 120 + x2 * (-1.0 / 5040 + x2 * (1.0 / 362880))))); }
__device__ __forceinline__ bool poison(double x) { return __double2hiint(x) == -1; }

// ---- persistent form --------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_persistent(Ctx c, double h) {
  extern __shared__ char lds_pad[];     // only there to set the number of workgroups a compute unit admits
  const int waves_per_member = c.nb / 16;
  const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (gw >= waves_per_member * c.members) return;
  const int m = gw / waves_per_member, w = gw % waves_per_member, ln = threadIdx.x & 63;
  const int b = w * 16 + (ln >> 2), k = ln & 3, kd = k < 3 ? k : 2;
  const int pb = neighbour(b, k, c.n1);
  const size_t mrec = (size_t)c.members * c.nb * 4, mvel = (size_t)c.members * c.nb * 3;
  // resident: parameters, step state, stage accelerations
  double par[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) par[e] = c.par[((size_t)m * c.nb * 4 + b * 4 + k) * 4 + e];
  LaneState S;
  S.invm = c.dofp[(((size_t)m * c.nb + b) * 3 + kd) * 2]; S.damp = c.dofp[(((size_t)m * c.nb + b) * 3 + kd) * 2 + 1];
#pragma unroll
  for (int l = 0; l < kStages; ++l) S.a[l] = 0.0;
  double o[4];
  {
    const double* r0 = c.rec + ((size_t)m * c.nb + b) * 4;     // ordinal 0: written by the host
    o[0] = r0[0]; o[1] = r0[1]; o[2] = r0[2]; o[3] = r0[3];
    S.q = r0[kd]; S.v = c.vel[((size_t)m * c.nb + b) * 3 + kd];
  }
  double v_i = S.v;
  const u32 off_p = (u32)(((size_t)m * c.nb + pb) * 32), off_o = (u32)(((size_t)m * c.nb + b) * 32 + 16 * (k & 1));
  const u32 off_v = (u32)((((size_t)m * c.nb + b) * 3 + kd) * 8);
  const int extra = c.skew && (gw % 7 == 3) ? c.skew : 0;
  for (int n = 0; n < c.steps; ++n) {
#pragma unroll
    for (int i = 0; i < kStages; ++i) {
      const size_t ord = (size_t)n * kStages + i;
      // the partner's record of this stage: poll until whole
      const double* base = c.ring ? c.ringbuf + (ord % c.ring) * mrec : c.rec + ord * mrec;
      double p[4];
      int spins = 0;
      for (;;) {
        v4u c0, c1;
        asm volatile("global_load_dwordx4 %0, %2, %3 sc1\n\tglobal_load_dwordx4 %1, %2, %3 offset:16 sc1\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(c0), "=&v"(c1) : "v"(off_p), "s"(base) : "memory");
        p[0] = __hiloint2double(c0.y, c0.x); p[1] = __hiloint2double(c0.w, c0.z);
        p[2] = __hiloint2double(c1.y, c1.x); p[3] = __hiloint2double(c1.w, c1.z);
        const bool ok = !(poison(p[0]) | poison(p[1]) | poison(p[2]) | poison(p[3]));
        if (__all(ok)) break;
        if (++spins > (1 << 22)) { c.err[0] = 1 + (int)ord; return; }
        if (!c.nosleep) __builtin_amdgcn_s_sleep(1);
      }
      double fx, fy, fth;
      ligament(o, p, par, c.iters + extra, fx, fy, fth);
      fx = quad_sum(fx); fy = quad_sum(fy); fth = quad_sum(fth);
      const double dE = k == 0 ? fx : (k == 1 ? fy : fth);
      double qnext, vnext;
      epilogue(S, i, dE, v_i, h, qnext, vnext);
      if (i == kStages - 1) { S.q = qnext; S.v = vnext; }
      v_i = vnext;
      // new record (x, y, th, sin th/2): every lane of the quad keeps a copy, lanes 0 and 1 publish one 16-byte chunk each
      o[0] = quad_bcast<0>(qnext); o[1] = quad_bcast<1>(qnext); o[2] = quad_bcast<2>(qnext);
      o[3] = half_sin(o[2]);
      double* nbase = c.ring ? c.ringbuf + ((ord + 1) % c.ring) * mrec : c.rec + (ord + 1) * mrec;
      if (c.selfpoison && k < 2 && ord + kAhead <= (size_t)c.steps * kStages) {
        // the owner poisons the place of a record it will write kAhead stages from now: that store has completed (the s_waitcnt in every
        // poll covers it) long before a neighbour can ask for that record -- a neighbour reaches stage t only after this wave has finished
        // stage t - 2.  The host poisons the first kAhead - 1 records of a launch only.
        v4u x; x.x = x.y = x.z = x.w = 0xFFFFFFFFu;
        asm volatile("global_store_dwordx4 %0, %1, %2 sc1\n\ts_nop 1" ::"v"(off_o), "v"(x), "s"(c.ring ? c.ringbuf + ((ord + kAhead) % c.ring) * mrec : c.rec + (ord + kAhead) * mrec) : "memory");
      }
      if (k < 2) {
        v4u x;
        const double s0 = k == 0 ? o[0] : o[2], s1 = k == 0 ? o[1] : o[3];
        x.x = __double2loint(s0); x.y = __double2hiint(s0); x.z = __double2loint(s1); x.w = __double2hiint(s1);
        asm volatile("global_store_dwordx4 %0, %1, %2 sc1\n\ts_nop 1" ::"v"(off_o), "v"(x), "s"(nbase) : "memory");
        if (c.ring) *reinterpret_cast<double2*>(reinterpret_cast<char*>(c.rec + (ord + 1) * mrec) + off_o) = make_double2(s0, s1);   // the checkpoint copy
      }
      if (k < 3) *reinterpret_cast<double*>(reinterpret_cast<char*>(c.vel + (ord + 1) * mvel) + off_v) = vnext;
    }
  }
}

// ---- baseline: one launch per stage, plain loads and stores -------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_stage(Ctx c, double h, int n, int i, double* A /* [member][6][block][3] */) {
  const int waves_per_member = c.nb / 16;
  const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (gw >= waves_per_member * c.members) return;
  const int m = gw / waves_per_member, w = gw % waves_per_member, ln = threadIdx.x & 63;
  const int b = w * 16 + (ln >> 2), k = ln & 3, kd = k < 3 ? k : 2;
  const int pb = neighbour(b, k, c.n1);
  const size_t mrec = (size_t)c.members * c.nb * 4, mvel = (size_t)c.members * c.nb * 3;
  const size_t ord = (size_t)n * kStages + i, ord0 = (size_t)n * kStages;
  double par[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) par[e] = c.par[((size_t)m * c.nb * 4 + b * 4 + k) * 4 + e];
  LaneState S;
  S.invm = c.dofp[(((size_t)m * c.nb + b) * 3 + kd) * 2]; S.damp = c.dofp[(((size_t)m * c.nb + b) * 3 + kd) * 2 + 1];
  double* Am = A + ((size_t)m * kStages * c.nb + b) * 3 + kd;
#pragma unroll
  for (int l = 0; l < kStages; ++l) S.a[l] = l < i ? Am[(size_t)l * c.nb * 3] : 0.0;
  const double* ro = c.rec + ord * mrec + ((size_t)m * c.nb + b) * 4;
  const double* rp = c.rec + ord * mrec + ((size_t)m * c.nb + pb) * 4;
  double o[4] = {ro[0], ro[1], ro[2], ro[3]}, p[4] = {rp[0], rp[1], rp[2], rp[3]};
  S.q = c.rec[ord0 * mrec + ((size_t)m * c.nb + b) * 4 + kd];
  S.v = c.vel[ord0 * mvel + ((size_t)m * c.nb + b) * 3 + kd];
  const double v_i = c.vel[ord * mvel + ((size_t)m * c.nb + b) * 3 + kd];
  const int extra = c.skew && (gw % 7 == 3) ? c.skew : 0;
  double fx, fy, fth;
  ligament(o, p, par, c.iters + extra, fx, fy, fth);
  fx = quad_sum(fx); fy = quad_sum(fy); fth = quad_sum(fth);
  const double dE = k == 0 ? fx : (k == 1 ? fy : fth);
  double qnext, vnext;
  epilogue(S, i, dE, v_i, h, qnext, vnext);
  if (k < 3) Am[(size_t)i * c.nb * 3] = (-dE - S.damp * v_i) * S.invm;
  o[0] = quad_bcast<0>(qnext); o[1] = quad_bcast<1>(qnext); o[2] = quad_bcast<2>(qnext);
  o[3] = half_sin(o[2]);
  double* no = c.rec + (ord + 1) * mrec + ((size_t)m * c.nb + b) * 4;
  if (k < 2) *reinterpret_cast<double2*>(no + 2 * k) = k == 0 ? make_double2(o[0], o[1]) : make_double2(o[2], o[3]);
  if (k < 3) c.vel[(ord + 1) * mvel + ((size_t)m * c.nb + b) * 3 + kd] = vnext;
}

__global__ void k_compare(const u64* a, const u64* b, size_t n, unsigned long long* diff) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long d = 0;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) d += a[i] != b[i];
  if (d) atomicAdd(diff, d);
}
__global__ void k_census(int* cu_count) {   // which compute unit did each workgroup land on?  (HW_REG_HW_ID: cu_id 8..11, sh 12, se 13..15; XCC_ID separate)
  if (threadIdx.x == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    atomicAdd(&cu_count[((xcc & 7) * 8 + se) * 32 + sh * 16 + cu], 1);
    __builtin_amdgcn_s_sleep(127);
  }
  __syncthreads();
}

static double half_sin_host(double th) { const double x = 0.5 * th, x2 = x * x; return x * (1.0 + x2 * (-1.0 / 6 + x2 * (1.0 / 120 + x2 * (-1.0 / 5040 + x2 * (1.0 / 362880))))); }

int main(int argc, char** argv) {
  const int members = argc > 1 ? atoi(argv[1]) : 1, steps = argc > 2 ? atoi(argv[2]) : 200, n1 = argc > 3 ? atoi(argv[3]) : 128;
  const int skew = argc > 4 ? atoi(argv[4]) : 0, iters = argc > 5 ? atoi(argv[5]) : 22, nosleep = argc > 6 ? atoi(argv[6]) : 0, selfpoison = argc > 7 ? atoi(argv[7]) : 0, ring = argc > 8 ? atoi(argv[8]) : 0;
  const int nb = n1 * n1;
  if (nb % 16) { printf("n1*n1 must be a multiple of 16\n"); return 1; }
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int n_cu = prop.multiProcessorCount;
  const size_t n_ord = (size_t)steps * kStages + 1;
  const size_t rec_elems = n_ord * members * nb * 4, vel_elems = n_ord * members * nb * 3;
  printf("members %d, %dx%d blocks, %d steps, %d CUs, records %.1f MB, skew %d, iters %d\n", members, n1, n1, steps, n_cu, rec_elems * 8 / 1e6, skew, iters);
  double *recP, *recB, *velP, *velB, *par, *dofp, *A;
  int* err; unsigned long long* diff;
  CK(hipMalloc(&recP, rec_elems * 8)); CK(hipMalloc(&recB, rec_elems * 8));
  CK(hipMalloc(&velP, vel_elems * 8)); CK(hipMalloc(&velB, vel_elems * 8));
  CK(hipMalloc(&par, (size_t)members * nb * 16 * 8)); CK(hipMalloc(&dofp, (size_t)members * nb * 6 * 8));
  CK(hipMalloc(&A, (size_t)members * kStages * nb * 3 * 8));
  CK(hipMalloc(&err, 64)); CK(hipMalloc(&diff, 8));
  // Dormand-Prince coefficients in acceleration form (values only need to be the same in both forms)
  {
    const double a[7][6] = {{0}, {1. / 5}, {3. / 40, 9. / 40}, {44. / 45, -56. / 15, 32. / 9}, {19372. / 6561, -25360. / 2187, 64448. / 6561, -212. / 729},
                            {9017. / 3168, -355. / 33, 46732. / 5247, 49. / 176, -5103. / 18656}, {35. / 384, 0, 500. / 1113, 125. / 192, -2187. / 6784, 11. / 84}};
    const double cc[7] = {0, 1. / 5, 3. / 10, 4. / 5, 8. / 9, 1, 1};
    double cv[6][6] = {}, cq[6][6] = {};
    for (int i = 0; i < 6; ++i)
      for (int l = 0; l <= i; ++l) {
        cv[i][l] = a[i + 1][l];
        double s = 0; for (int j = l + 1; j <= i; ++j) s += a[i + 1][j] * a[j][l];
        cq[i][l] = s;
      }
    CK(hipMemcpyToSymbol(HIP_SYMBOL(c_cv), cv, sizeof(cv))); CK(hipMemcpyToSymbol(HIP_SYMBOL(c_cq), cq, sizeof(cq))); CK(hipMemcpyToSymbol(HIP_SYMBOL(c_c), cc, sizeof(cc)));
  }
  std::vector<double> hp((size_t)members * nb * 16), hd((size_t)members * nb * 6), hr((size_t)members * nb * 4), hv((size_t)members * nb * 3);
  unsigned long long s = 88172645463325252ull;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) / 9007199254740992.0; };
  for (auto& x : hp) x = rnd() - 0.5;
  for (size_t i = 0; i < hd.size(); i += 2) { hd[i] = 0.5 + rnd(); hd[i + 1] = 0.1 * rnd(); }
  for (size_t i = 0; i < hr.size(); i += 4) { hr[i] = 0.1 * rnd(); hr[i + 1] = 0.1 * rnd(); hr[i + 2] = 0.2 * rnd() - 0.1; hr[i + 3] = sin(0.5 * hr[i + 2]); }
  for (auto& x : hv) x = 0.01 * (rnd() - 0.5);
  CK(hipMemcpy(par, hp.data(), hp.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dofp, hd.data(), hd.size() * 8, hipMemcpyHostToDevice));
  double* ringbuf = nullptr;
  if (ring) CK(hipMalloc(&ringbuf, (size_t)ring * members * nb * 4 * 8));
  Ctx c{n1, nb, members, steps, iters, skew, nosleep, selfpoison, ring, ringbuf, recP, velP, par, dofp, err};
  const double h = 1e-3;
  const int waves = members * nb / 16, grid = (waves + 3) / 4;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms;
  // ---- persistent
  const int wg_per_cu = (grid + n_cu - 1) / n_cu;
  if (wg_per_cu > 8) { printf("too many workgroups to be resident (%d per CU)\n", wg_per_cu); return 1; }
  int lds = wg_per_cu == 1 ? 96 * 1024 : (160 * 1024 / wg_per_cu) & ~1023;   // admits exactly wg_per_cu workgroups per compute unit
  if (getenv("MOCK_LDS")) lds = atoi(getenv("MOCK_LDS"));     // (experiment: does the dispatcher spread workgroups by itself?)
  CK(hipFuncSetAttribute((const void*)k_persistent, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  {
    int* cu_count; CK(hipMalloc(&cu_count, 8 * 8 * 32 * 4)); CK(hipMemset(cu_count, 0, 8 * 8 * 32 * 4));
    CK(hipFuncSetAttribute((const void*)k_census, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL(k_census, dim3(grid), dim3(256), lds, 0, cu_count);
    std::vector<int> hc(8 * 8 * 32);
    CK(hipMemcpy(hc.data(), cu_count, hc.size() * 4, hipMemcpyDeviceToHost));
    int used = 0, mx = 0; for (int x : hc) { used += x > 0; mx = x > mx ? x : mx; }
    printf("census: %d workgroups (%d B LDS each) on %d distinct CUs, at most %d per CU\n", grid, lds, used, mx);
  }
  for (int rep = 0; rep < 3; ++rep) {
    // every repetition starts from other data: a stale copy of an earlier repetition's records, wherever it may sit, is then a WRONG value
    for (size_t i = 0; i < hr.size(); i += 4) { hr[i] *= 0.9; hr[i + 1] *= 1.1; hr[i + 2] *= 0.95; hr[i + 3] = half_sin_host(hr[i + 2]); }
    if (selfpoison) {       // finite garbage everywhere except the first kAhead - 1 records: a missing poison shows as a wrong result, not as a hang
      std::vector<double> junk(1 << 20, 1.0 + rep);
      for (size_t o = 0; o < rec_elems; o += junk.size()) CK(hipMemcpy(recP + o, junk.data(), std::min(junk.size(), rec_elems - o) * 8, hipMemcpyHostToDevice));
      CK(hipMemset(recP + (size_t)members * nb * 4, 0xFF, (size_t)(kAhead - 1) * members * nb * 4 * 8));
    } else CK(hipMemset(recP, 0xFF, rec_elems * 8));
    CK(hipMemset(velP, 0xFF, vel_elems * 8)); CK(hipMemset(err, 0, 64));
    CK(hipMemcpy(recP, hr.data(), hr.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(velP, hv.data(), hv.size() * 8, hipMemcpyHostToDevice));
    if (ring) {
      std::vector<double> junk((size_t)ring * members * nb * 4, 2.0 + rep);
      CK(hipMemcpy(ringbuf, junk.data(), junk.size() * 8, hipMemcpyHostToDevice));
      CK(hipMemcpy(ringbuf, hr.data(), hr.size() * 8, hipMemcpyHostToDevice));
      CK(hipMemset(ringbuf + (size_t)members * nb * 4, 0xFF, (size_t)(kAhead - 1) * members * nb * 4 * 8));
    }
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_persistent, dim3(grid), dim3(256), lds, 0, c, h);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    int herr; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    printf("persistent           : %8.3f ms = %6.3f us per stage   (give-up code %d)\n", ms, ms * 1e3 / (steps * kStages), herr);
  }
  c.rec = recB; c.vel = velB;
  // ---- baseline, from the data of the last repetition
  CK(hipMemset(recB, 0xFF, rec_elems * 8)); CK(hipMemset(velB, 0xFF, vel_elems * 8));
  CK(hipMemcpy(recB, hr.data(), hr.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(velB, hv.data(), hv.size() * 8, hipMemcpyHostToDevice));
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipEventRecord(e0));
    for (int n = 0; n < steps; ++n)
      for (int i = 0; i < kStages; ++i) hipLaunchKernelGGL(k_stage, dim3(grid), dim3(256), 0, 0, c, h, n, i, A);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
  }
  printf("one launch per stage : %8.3f ms = %6.3f us per stage\n", ms, ms * 1e3 / (steps * kStages));
  CK(hipMemset(diff, 0, 8));
  hipLaunchKernelGGL(k_compare, dim3(2048), dim3(256), 0, 0, (const u64*)recP, (const u64*)recB, rec_elems, diff);
  hipLaunchKernelGGL(k_compare, dim3(2048), dim3(256), 0, 0, (const u64*)velP, (const u64*)velB, vel_elems, diff);
  unsigned long long hdiff; CK(hipMemcpy(&hdiff, diff, 8, hipMemcpyDeviceToHost));
  std::vector<double> last(4); CK(hipMemcpy(last.data(), recP + (n_ord - 1) * members * nb * 4, 32, hipMemcpyDeviceToHost));
  printf("words that differ between the two forms: %llu of %zu   (last record of block 0: %.6g %.6g %.6g %.6g)\n", hdiff, rec_elems + vel_elems, last[0], last[1], last[2], last[3]);
  return hdiff != 0;
}
