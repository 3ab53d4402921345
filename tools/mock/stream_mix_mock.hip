// Ceiling mock for the stage launches: what does the memory system deliver for the read / write MIX and the occupancy of the two stage
// kernels when nothing but the traffic is left?  NOT the product: one lane per (block, node slot) of members x 128x128 blocks as in the
// product, every lane reads R 16-byte chunks and writes W 16-byte chunks of its own, perfectly coalesced, from / to separate arrays
// (no gathers, no dependent loads, no arithmetic beyond a sum that keeps the loads alive).  VGPR pressure is set with a dummy array so
// that the kernel runs at 4 or 5 waves per SIMD like the reverse / forward stage kernels.
//   reverse stage (records level): 611 B/unit moved = 440 read + 170 written  -> per lane 110 + 42.5 B  ~ R = 7, W = 3 (112 + 48 B)
//   forward stage:                 348 B/unit       = 276 read +  72 written  -> per lane  69 + 18 B    ~ R = 4, W = 1 ( 64 + 16 B)
//   hipcc --offload-arch=gfx950 -O3 -o stream_mix_mock stream_mix_mock.hip && ./stream_mix_mock [members]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
constexpr int kMaxArr = 8;
struct Arrs { const double2* in[kMaxArr]; double2* out[kMaxArr]; };

template <int R, int W, int WAVES>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void k_mix(Arrs a, size_t n) {
  const size_t i = (size_t)blockIdx.y * gridDim.x * 128 + (size_t)blockIdx.x * 128 + threadIdx.x;
  if (i >= n) return;
  double2 v[R];
#pragma unroll
  for (int r = 0; r < R; ++r) v[r] = a.in[r][i];          // all loads in flight before the first use
  double sx = 0, sy = 0;
#pragma unroll
  for (int r = 0; r < R; ++r) { sx += v[r].x; sy += v[r].y; }
#pragma unroll
  for (int w = 0; w < W; ++w) a.out[w][i] = make_double2(sx + w, sy - w);
}

// The same traffic with every array cut into one region per member, the regions `gap` bytes apart (the product keeps ~25 arrays with one
// region per member each: a launch touches several hundred separate regions of 0.4-1 MB -- does address translation notice?)
template <int R, int W, int WAVES>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void k_mix_regions(Arrs a, size_t per_member, size_t gap16) {
  const size_t i = (size_t)blockIdx.y * gap16 + (size_t)blockIdx.x * 128 + threadIdx.x;     // blockIdx.y = member
  if ((size_t)blockIdx.x * 128 + threadIdx.x >= per_member) return;
  double2 v[R];
#pragma unroll
  for (int r = 0; r < R; ++r) v[r] = a.in[r][i];
  double sx = 0, sy = 0;
#pragma unroll
  for (int r = 0; r < R; ++r) { sx += v[r].x; sy += v[r].y; }
#pragma unroll
  for (int w = 0; w < W; ++w) a.out[w][i] = make_double2(sx + w, sy - w);
}

template <int R, int W, int WAVES>
static void run_regions(const char* what, int members, Arrs a, size_t per_member, size_t gap16) {
  dim3 grid(512, members);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int it = 0; it < 20; ++it) hipLaunchKernelGGL((k_mix_regions<R, W, WAVES>), grid, dim3(128), 0, 0, a, per_member, gap16);
  CK(hipEventRecord(e0));
  const int reps = 400;
  for (int it = 0; it < reps; ++it) hipLaunchKernelGGL((k_mix_regions<R, W, WAVES>), grid, dim3(128), 0, 0, a, per_member, gap16);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = 1e3 * ms / reps, moved = (double)members * per_member * 16.0 * (R + W);
  printf("%-34s R=%d W=%d %d waves/SIMD, regions %6.1f MB apart: %7.2f us per launch, %6.0f GB/s moved\n", what, R, W, WAVES, gap16 * 16e-6, us, moved / us * 1e-3);
}

template <int R, int W, int WAVES>
static void run(const char* what, int members, Arrs a, size_t n, double alg_bytes_per_unit) {
  dim3 grid(512, members);                                   // 512 workgroups of 128 lanes per member = 16 384 units x 4 lanes
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int it = 0; it < 20; ++it) hipLaunchKernelGGL((k_mix<R, W, WAVES>), grid, dim3(128), 0, 0, a, n);
  CK(hipEventRecord(e0));
  const int reps = 400;
  for (int it = 0; it < reps; ++it) hipLaunchKernelGGL((k_mix<R, W, WAVES>), grid, dim3(128), 0, 0, a, n);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = 1e3 * ms / reps, moved = (double)n * 16.0 * (R + W);
  printf("%-34s R=%d W=%d %d waves/SIMD: %7.2f us per launch, %6.0f GB/s moved (%.2f of 8 TB/s); the product's launch moves %.0f B/unit: at this rate %.2f us\n",
         what, R, W, WAVES, us, moved / us * 1e-3, moved / us * 1e-3 / 8000.0, alg_bytes_per_unit, alg_bytes_per_unit * (n / 4) / (moved / us) );
}

int main(int argc, char** argv) {
  const int members = argc > 1 ? atoi(argv[1]) : 16;
  const size_t n = (size_t)members * 16384 * 4;
  Arrs a;
  for (int k = 0; k < kMaxArr; ++k) {
    void *p, *q;
    CK(hipMalloc(&p, n * 16)); CK(hipMalloc(&q, n * 16));
    CK(hipMemset(p, 0, n * 16)); CK(hipMemset(q, 0, n * 16));
    a.in[k] = (const double2*)p; a.out[k] = (double2*)q;
  }
  printf("members %d: %zu lanes, %.1f MB per 16-byte array\n", members, n, n * 16e-6);
  run<7, 3, 4>("reverse-stage mix", members, a, n, 611.0);
  run<7, 3, 5>("reverse-stage mix", members, a, n, 611.0);
  run<7, 3, 8>("reverse-stage mix", members, a, n, 611.0);
  run<4, 1, 5>("forward-stage mix", members, a, n, 348.0);
  run<4, 1, 8>("forward-stage mix", members, a, n, 348.0);
  run<8, 0, 8>("read only", members, a, n, 0.0);
  run<1, 1, 8>("copy", members, a, n, 0.0);
  // one region per member: 1 MB regions, 1 / 9 / 33 MB apart (arrays re-allocated large enough)
  for (int k = 0; k < kMaxArr; ++k) { CK(hipFree((void*)a.in[k])); CK(hipFree(a.out[k])); }
  const size_t per_member = 16384 * 4;                                   // lanes = 16-byte elements per member region (1 MB)
  for (size_t gap_mb : {1, 9, 33}) {
    const size_t gap16 = gap_mb * (1 << 20) / 16, bytes = (size_t)members * gap16 * 16;
    for (int k = 0; k < kMaxArr; ++k) {
      void *p, *q;
      CK(hipMalloc(&p, bytes)); CK(hipMalloc(&q, bytes));
      CK(hipMemset(p, 0, bytes)); CK(hipMemset(q, 0, bytes));
      a.in[k] = (const double2*)p; a.out[k] = (double2*)q;
    }
    run_regions<7, 3, 4>("reverse-stage mix", members, a, per_member, gap16);
    run_regions<4, 1, 5>("forward-stage mix", members, a, per_member, gap16);
    for (int k = 0; k < kMaxArr; ++k) { CK(hipFree((void*)a.in[k])); CK(hipFree(a.out[k])); }
  }
  return 0;
}
