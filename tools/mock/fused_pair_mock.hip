// Traffic mock for the structural step DESIGN.md section 6 proposes: two Runge-Kutta stages per launch on 16x16 lattice tiles with a
// one-ring halo, against two launches of one stage each.  NOT the product: synthetic arrays with the forward stage kernel's access
// pattern (one lane per (block, node slot); own 32-B record by lanes 0-1, velocity / inverse mass / stage accelerations by lanes 0-2,
// 16-B node vector and slot index per lane, partner record + partner node vector gathered through the index) and ~60 dependent FMAs
// per lane; the question it answers is what the memory system makes of the fused access pattern, nothing about physics.
//   hipcc --offload-arch=gfx950 -O3 -o fused_pair_mock fused_pair_mock.hip && ./fused_pair_mock [members]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
constexpr int N = 128, NB = N * N, T = 16;             // lattice side, blocks, tile side
struct Arr { double *rec, *vel, *pr, *invm, *A, *rec2, *vel2, *rec3, *vel3; int* idx; };

__device__ __forceinline__ double work(double a, double b, double c, double d) {
  double x = a;
#pragma unroll
  for (int k = 0; k < 30; ++k) x = fma(x, b, c) * 0.999 + d;
  return x;
}
// partner block of slot k of block (r, c): right / up / left / down, clamped
__device__ __forceinline__ int partner(int r, int c, int k) {
  int rr = r + (k == 1) - (k == 3), cc = c + (k == 0) - (k == 2);
  rr = min(max(rr, 0), N - 1); cc = min(max(cc, 0), N - 1);
  return rr * N + cc;
}
// one stage, one lane per slot (the product's shape); stage ordinal i = number of earlier accelerations read
__global__ __launch_bounds__(128) void k_stage(Arr a, int i, const double* rin, const double* vin, double* rout, double* vout) {
  const int m = blockIdx.y, slot = blockIdx.x * 128 + threadIdx.x, b = slot >> 2, k = slot & 3, kd = k < 3 ? k : 2;
  const size_t mo = (size_t)m * NB;
  const double2 pc = k < 2 ? *(const double2*)(rin + (mo + b) * 4 + 2 * k) : make_double2(0, 0);
  const double v = vin[(mo + b) * 3 + kd], im = a.invm[(mo + b) * 3 + kd];
  const double2 ro = *(const double2*)(a.pr + (mo * 4 + slot) * 2);
  const int pb = a.idx[slot];
  const double2 p0 = *(const double2*)(rin + (mo + pb) * 4), p1 = *(const double2*)(rin + (mo + pb) * 4 + 2);
  const double2 rp = *(const double2*)(a.pr + (mo * 4 + pb * 4 + ((k + 2) & 3)) * 2);
  double acc = 0;
  for (int l = 0; l < i; ++l) acc += a.A[((size_t)l * gridDim.y * NB + mo + b) * 3 + kd];
  double f = work(pc.x + p0.x + ro.x, 1.0000001, p1.y + rp.y, pc.y + p0.y + ro.y + p1.x + rp.x);
  f += __shfl_xor(f, 1) + __shfl_xor(f, 2);
  const double an = (f - v) * im + acc;
  if (k < 3) { a.A[((size_t)i * gridDim.y * NB + mo + b) * 3 + kd] = an; vout[(mo + b) * 3 + kd] = v + 1e-6 * an; }
  const double q = __shfl(an, (threadIdx.x & ~3) + 1) * 1e-6;
  if (k < 2) *(double2*)(rout + (mo + b) * 4 + 2 * k) = make_double2(pc.x + q, pc.y + 1e-6 * an);
}
// two stages per launch: a 1024-thread workgroup owns a 16x16 tile; phase A evaluates stage i on the tile and on its 64 halo blocks
// (their partners gathered from memory), leaves the stage-(i+1) records of tile + halo in LDS (and writes the tile's to memory: the
// records checkpoint wants them); phase B evaluates stage i+1 on the tile from LDS.
__global__ __launch_bounds__(1024) void k_pair(Arr a, int i, const double* rin, const double* vin, double* rmid, double* vmid, double* rout, double* vout) {
  __shared__ double lrec[(T + 2) * (T + 2)][4];
  __shared__ double lvel[T * T][3];
  const int m = blockIdx.y, tr = (blockIdx.x / (N / T)) * T, tc = (blockIdx.x % (N / T)) * T;
  const size_t mo = (size_t)m * NB;
  for (int pass = 0; pass < 2; ++pass) {
    int lr, lc, k;
    if (pass == 0) { const int lb = threadIdx.x >> 2; k = threadIdx.x & 3; lr = lb / T; lc = lb % T; }
    else {                                               // 64 halo blocks x 4 slots = 256 lanes
      if (threadIdx.x >= 256) break;
      const int hb = threadIdx.x >> 2; k = threadIdx.x & 3;
      const int side = hb / T, t = hb % T;
      lr = side == 0 ? -1 : (side == 1 ? T : t); lc = side == 2 ? -1 : (side == 3 ? T : t);
      if (side < 2) lc = t;
    }
    const int r = min(max(tr + lr, 0), N - 1), c = min(max(tc + lc, 0), N - 1), b = r * N + c, kd = k < 3 ? k : 2;
    const double2 pc = k < 2 ? *(const double2*)(rin + (mo + b) * 4 + 2 * k) : make_double2(0, 0);
    const double v = vin[(mo + b) * 3 + kd], im = a.invm[(mo + b) * 3 + kd];
    const double2 ro = *(const double2*)(a.pr + (mo * 4 + b * 4 + k) * 2);
    const int pb = partner(r, c, k);
    const double2 p0 = *(const double2*)(rin + (mo + pb) * 4), p1 = *(const double2*)(rin + (mo + pb) * 4 + 2);
    const double2 rp = *(const double2*)(a.pr + (mo * 4 + pb * 4 + ((k + 2) & 3)) * 2);
    double acc = 0;
    for (int l = 0; l < i; ++l) acc += a.A[((size_t)l * gridDim.y * NB + mo + b) * 3 + kd];
    double f = work(pc.x + p0.x + ro.x, 1.0000001, p1.y + rp.y, pc.y + p0.y + ro.y + p1.x + rp.x);
    f += __shfl_xor(f, 1) + __shfl_xor(f, 2);
    const double an = (f - v) * im + acc;
    const double q = __shfl(an, (threadIdx.x & ~3) + 1) * 1e-6;
    const int li = (lr + 1) * (T + 2) + (lc + 1);
    if (k < 2) { lrec[li][2 * k] = pc.x + q; lrec[li][2 * k + 1] = pc.y + 1e-6 * an; }
    if (pass == 0) {
      if (k < 3) { a.A[((size_t)i * gridDim.y * NB + mo + b) * 3 + kd] = an; lvel[lr * T + lc][kd] = v + 1e-6 * an; vmid[(mo + b) * 3 + kd] = v + 1e-6 * an; }
      if (k < 2) *(double2*)(rmid + (mo + b) * 4 + 2 * k) = make_double2(pc.x + q, pc.y + 1e-6 * an);
    }
  }
  __syncthreads();
  {                                                      // phase B: stage i + 1 on the tile, records from LDS
    const int lb = threadIdx.x >> 2, k = threadIdx.x & 3, lr = lb / T, lc = lb % T, kd = k < 3 ? k : 2;
    const int r = tr + lr, c = tc + lc, b = r * N + c;
    const int li = (lr + 1) * (T + 2) + (lc + 1), lp = (lr + 1 + (k == 1) - (k == 3)) * (T + 2) + (lc + 1 + (k == 0) - (k == 2));
    const double2 pc = k < 2 ? make_double2(lrec[li][2 * k], lrec[li][2 * k + 1]) : make_double2(0, 0);
    const double v = lvel[lr * T + lc][kd], im = a.invm[(mo + b) * 3 + kd];
    const double2 ro = *(const double2*)(a.pr + (mo * 4 + b * 4 + k) * 2);
    const int pb = partner(r, c, k);
    const double2 p0 = make_double2(lrec[lp][0], lrec[lp][1]), p1 = make_double2(lrec[lp][2], lrec[lp][3]);
    const double2 rp = *(const double2*)(a.pr + (mo * 4 + pb * 4 + ((k + 2) & 3)) * 2);
    double acc = 0;
    for (int l = 0; l < i + 1; ++l) acc += a.A[((size_t)l * gridDim.y * NB + mo + b) * 3 + kd];
    double f = work(pc.x + p0.x + ro.x, 1.0000001, p1.y + rp.y, pc.y + p0.y + ro.y + p1.x + rp.x);
    f += __shfl_xor(f, 1) + __shfl_xor(f, 2);
    const double an = (f - v) * im + acc;
    if (k < 3) { a.A[((size_t)(i + 1) * gridDim.y * NB + mo + b) * 3 + kd] = an; vout[(mo + b) * 3 + kd] = v + 1e-6 * an; }
    const double q = __shfl(an, (threadIdx.x & ~3) + 1) * 1e-6;
    if (k < 2) *(double2*)(rout + (mo + b) * 4 + 2 * k) = make_double2(pc.x + q, pc.y + 1e-6 * an);
  }
}
int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 16, reps = 300;
  Arr a;
  const size_t nb = (size_t)M * NB;
  CK(hipMalloc(&a.rec, nb * 32)); CK(hipMalloc(&a.rec2, nb * 32)); CK(hipMalloc(&a.rec3, nb * 32));
  CK(hipMalloc(&a.vel, nb * 24)); CK(hipMalloc(&a.vel2, nb * 24)); CK(hipMalloc(&a.vel3, nb * 24));
  CK(hipMalloc(&a.pr, nb * 64)); CK(hipMalloc(&a.invm, nb * 24)); CK(hipMalloc(&a.A, nb * 24 * 7)); CK(hipMalloc(&a.idx, NB * 4 * sizeof(int)));
  CK(hipMemset(a.rec, 0, nb * 32)); CK(hipMemset(a.vel, 0, nb * 24)); CK(hipMemset(a.pr, 0, nb * 64)); CK(hipMemset(a.A, 0, nb * 24 * 7));
  std::vector<double> one(nb * 3, 1.0); CK(hipMemcpy(a.invm, one.data(), nb * 24, hipMemcpyHostToDevice));
  std::vector<int> idx(NB * 4);
  for (int r = 0; r < N; ++r) for (int c = 0; c < N; ++c) for (int k = 0; k < 4; ++k) {
    int rr = std::min(std::max(r + (k == 1) - (k == 3), 0), N - 1), cc = std::min(std::max(c + (k == 0) - (k == 2), 0), N - 1);
    idx[(r * N + c) * 4 + k] = rr * N + cc;
  }
  CK(hipMemcpy(a.idx, idx.data(), idx.size() * sizeof(int), hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const dim3 gu(NB * 4 / 128, M), gf((N / T) * (N / T), M);
  float ms;
  for (int variant = 0; variant < 2; ++variant)
    for (int i = 1; i <= 3; i += 2) {
      for (int rep = -20; rep < reps; ++rep) {
        if (rep == 0) CK(hipEventRecord(e0));
        if (variant == 0) {
          hipLaunchKernelGGL(k_stage, gu, dim3(128), 0, 0, a, i, a.rec, a.vel, a.rec2, a.vel2);
          hipLaunchKernelGGL(k_stage, gu, dim3(128), 0, 0, a, i + 1, a.rec2, a.vel2, a.rec3, a.vel3);
        } else hipLaunchKernelGGL(k_pair, gf, dim3(1024), 0, 0, a, i, a.rec, a.vel, a.rec2, a.vel2, a.rec3, a.vel3);
      }
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
      printf("%s, stages %d and %d, %d members: %.2f us per pair of stages\n", variant ? "one fused launch (16x16 tiles + halo, LDS)" : "two launches                             ", i, i + 1, M, 1e3 * ms / reps);
    }
  return 0;
}
