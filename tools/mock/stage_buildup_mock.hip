// Build-up mock for the reverse stage launch: start from the traffic-only kernel of stream_mix_mock.hip (7 coalesced 16-byte reads + 3
// writes per lane, 24 us per 16-member launch) and add the product's features one at a time until the shape of k_adj_stage is reached
// (33 us): which one costs what?  NOT the product -- synthetic arrays and synthetic arithmetic.  Lane = (block, node slot) of members x
// 128x128 blocks; 4 waves per SIMD throughout (the reverse kernel's occupancy).
//   G  gathers      : 3 of the reads become gathers from the neighbouring block of the lane's slot (right / up / left / down), i.e.
//                     bytes other lanes also read (served by L2), plus 2 more gathered reads (the partner's w)
//   N  narrow       : 3 of the coalesced 16-byte reads become 6 8-byte reads of per-DOF arrays (3 doubles per block, lane 3 repeats lane 2)
//   A  arithmetic   : ~600 vector instructions between the loads and the stores (fp64 FMA chains, 8 independent chains)
//   P  two phases   : 2 of the reads (the accumulators' old values) are issued only after the arithmetic and written back to the same
//                     addresses (read-modify-write), as the reverse stage does
//   hipcc --offload-arch=gfx950 -O3 -o stage_buildup_mock stage_buildup_mock.hip && ./stage_buildup_mock [members]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
constexpr int N1 = 128, NB = N1 * N1;
struct Arrs { const double2* in[8]; double2* out[4]; const double* dof[6]; double2* acc[2]; };

__device__ __forceinline__ int neighbour(int b, int k) {
  int r = b / N1, c = b % N1;
  r += (k == 1) - (k == 3); c += (k == 0) - (k == 2);
  r = min(max(r, 0), N1 - 1); c = min(max(c, 0), N1 - 1);
  return r * N1 + c;
}

template <int G, int NARROW, int A, int P>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_stage(Arrs a) {
  const int m = blockIdx.y, lane = blockIdx.x * 128 + threadIdx.x, b = lane >> 2, k = lane & 3;
  const size_t base = (size_t)m * NB * 4, i = base + lane;
  double2 v[7];
  // reads 0-2: own data, coalesced 16 B (or, NARROW: six 8-byte per-DOF reads)
  if (NARROW) {
    const size_t d = ((size_t)m * NB + b) * 3 + (k < 3 ? k : 2);
#pragma unroll
    for (int r = 0; r < 3; ++r) v[r] = make_double2(a.dof[2 * r][d], a.dof[2 * r + 1][d]);
  } else {
#pragma unroll
    for (int r = 0; r < 3; ++r) v[r] = a.in[r][i];
  }
  // reads 3-4: own data, coalesced 16 B -- unless P: then they are the accumulators, read after the arithmetic
  if (!P) { v[3] = a.in[3][i]; v[4] = a.in[4][i]; } else { v[3] = v[0]; v[4] = v[1]; }
  // reads 5-6 (+ 3 more with G): coalesced, or gathered from the neighbouring block
  double2 g0 = make_double2(0, 0), g1 = g0, g2 = g0;
  if (G == 1) {
    const size_t j = base + (size_t)neighbour(b, k) * 4;
    v[5] = a.in[5][j]; v[6] = a.in[5][j + 1]; g0 = a.in[6][j + ((k + 2) & 3)]; g1 = a.in[7][j]; g2 = a.in[7][j + 1];
  } else if (G == 2) {
    // a wave = a 4 x 4 tile of blocks (tile-major records): the neighbour's data sits in another lane of the wave for 3 of 4 blocks per
    // direction (ds_bpermute), only the tile's rim gathers from memory
    const int tb = (threadIdx.x & 63) >> 2, tr = tb >> 2, tc = tb & 3;
    const int dr = (k == 1) - (k == 3), dc = (k == 0) - (k == 2);
    const bool inside = (unsigned)(tr + dr) < 4u && (unsigned)(tc + dc) < 4u;
    const int src = ((tr + dr) * 4 + (tc + dc)) * 4;                    // first lane of the neighbour's quad
    const double2 o5 = a.in[5][i], o6 = a.in[6][i], o7 = a.in[7][i];   // own data, coalesced
    auto sh = [&](double x, int l) { return __shfl(x, l, 64); };
    v[5] = make_double2(sh(o5.x, src), sh(o5.y, src)); v[6] = make_double2(sh(o5.x, src + 1), sh(o5.y, src + 1));
    g0 = make_double2(sh(o6.x, src + ((k + 2) & 3)), sh(o6.y, src + ((k + 2) & 3)));
    g1 = make_double2(sh(o7.x, src), sh(o7.y, src)); g2 = make_double2(sh(o7.x, src + 1), sh(o7.y, src + 1));
    if (!inside) {
      const size_t j = base + (size_t)neighbour(b, k) * 4;
      v[5] = a.in[5][j]; v[6] = a.in[5][j + 1]; g0 = a.in[6][j + ((k + 2) & 3)]; g1 = a.in[7][j]; g2 = a.in[7][j + 1];
    }
  } else { v[5] = a.in[5][i]; v[6] = a.in[6][i]; }
  double x[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) x[c] = v[c % 7].x + v[(c + 3) % 7].y + g0.x + g1.y + g2.x;
  if (A) {
#pragma unroll 1
    for (int it = 0; it < A; ++it) {
#pragma unroll
      for (int c = 0; c < 8; ++c) x[c] = fma(x[c], 1.0000001, x[(c + 1) & 7] * 1e-9);
    }
  }
  double sx = 0, sy = 0;
#pragma unroll
  for (int c = 0; c < 8; ++c) { sx += x[c]; sy -= x[c]; }
  if (P) {
    const double2 o0 = a.acc[0][i], o1 = a.acc[1][i];
    a.acc[0][i] = make_double2(o0.x + 1e-9 * sx, o0.y + 1e-9 * sy);
    a.acc[1][i] = make_double2(o1.x - 1e-9 * sx, o1.y - 1e-9 * sy);
    a.out[0][i] = make_double2(sx, sy);
  } else {
#pragma unroll
    for (int w = 0; w < 3; ++w) a.out[w][i] = make_double2(sx + w, sy - w);
  }
}

template <int G, int NARROW, int A, int P>
static void run(const char* what, int members, Arrs a) {
  dim3 grid(512, members);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int it = 0; it < 20; ++it) hipLaunchKernelGGL((k_stage<G, NARROW, A, P>), grid, dim3(128), 0, 0, a);
  CK(hipEventRecord(e0));
  const int reps = 300;
  for (int it = 0; it < reps; ++it) hipLaunchKernelGGL((k_stage<G, NARROW, A, P>), grid, dim3(128), 0, 0, a);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-64s %7.2f us per launch\n", what, 1e3 * ms / reps);
}

int main(int argc, char** argv) {
  const int members = argc > 1 ? atoi(argv[1]) : 16;
  const size_t n = (size_t)members * NB * 4;
  Arrs a;
  auto alloc = [&](size_t bytes) { void* p; CK(hipMalloc(&p, bytes)); CK(hipMemset(p, 0, bytes)); return p; };
  for (int k = 0; k < 8; ++k) a.in[k] = (const double2*)alloc(n * 16);
  for (int k = 0; k < 4; ++k) a.out[k] = (double2*)alloc(n * 16);
  for (int k = 0; k < 6; ++k) a.dof[k] = (const double*)alloc((size_t)members * NB * 3 * 8);
  for (int k = 0; k < 2; ++k) a.acc[k] = (double2*)alloc(n * 16);
  printf("members %d (4 waves per SIMD; 75 loop iterations of 8 FMAs + 8 multiplies ~ 600 fp64 instructions per lane)\n", members);
  run<0, 0, 0, 0>("traffic only: 7 coalesced 16-B reads + 3 writes", members, a);
  run<1, 0, 0, 0>("+ gathers (5 reads from the neighbouring block)", members, a);
  run<0, 1, 0, 0>("+ narrow (six 8-B per-DOF reads instead of three 16-B)", members, a);
  run<0, 0, 38, 0>("+ arithmetic (~600 fp64 instructions)", members, a);
  run<0, 0, 0, 1>("+ two phases (accumulators read after the arithmetic, RMW)", members, a);
  run<1, 1, 0, 0>("gathers + narrow", members, a);
  run<1, 1, 38, 0>("gathers + narrow + arithmetic", members, a);
  run<1, 1, 0, 1>("gathers + narrow + two phases", members, a);
  run<1, 1, 38, 1>("gathers + narrow + arithmetic + two phases (the stage's shape)", members, a);
  run<1, 1, 76, 1>("  ... with twice the arithmetic", members, a);
  run<0, 0, 38, 1>("arithmetic + two phases, no gathers, wide reads", members, a);
  run<2, 1, 0, 1>("tile-wave exchange (3 of 4 neighbours by ds_bpermute) + narrow + two phases", members, a);
  run<2, 1, 38, 1>("tile-wave exchange + narrow + arithmetic + two phases", members, a);
  run<2, 1, 19, 1>("tile-wave exchange + narrow + half the arithmetic + two phases", members, a);
  run<0, 0, 19, 0>("traffic + HALF the arithmetic (~300 fp64 instructions)", members, a);
  run<1, 1, 19, 1>("the stage's shape with half the arithmetic", members, a);
  return 0;
}
