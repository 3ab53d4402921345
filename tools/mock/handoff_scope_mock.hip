// Round 6 question: what does ONE hand-off of the persistent stage loop cost by cache scope?  The loop hands a 32-byte record from the wave that
// owns a block to its neighbour waves with sc1 stores and sc1 loads (device scope: past the L1 AND coherent across the eight XCDs' L2s),
// measured ~1.5 us per stage.  Waves on the SAME XCD share an L2: a plain store (write-through L1 -> L2) and an sc0 load (misses the L1, may hit
// the L2) would be enough -- if all waves of a member sit on one XCD.  This mock times a ping-pong between two workgroups by scope and by
// placement, and prints every workgroup's XCC_ID (is workgroup w really on XCD w % 8?).   NOT the product.
//   hipcc --offload-arch=gfx950 -O3 -o handoff_scope_mock handoff_scope_mock.hip && ./handoff_scope_mock [rounds=2000]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef unsigned v4u __attribute__((ext_vector_type(4)));

template <int LD>
__device__ __forceinline__ v4u ld16(const void* base, unsigned off) {
  v4u x;
  if (LD == 0) asm volatile("global_load_dwordx4 %0, %1, %2 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(x) : "v"(off), "s"(base) : "memory");
  if (LD == 1) asm volatile("global_load_dwordx4 %0, %1, %2 sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(x) : "v"(off), "s"(base) : "memory");
  if (LD == 2) asm volatile("global_load_dwordx4 %0, %1, %2 sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(x) : "v"(off), "s"(base) : "memory");
  if (LD == 3) asm volatile("global_load_dwordx4 %0, %1, %2 nt\n\ts_waitcnt vmcnt(0)" : "=&v"(x) : "v"(off), "s"(base) : "memory");
  return x;
}
template <int ST>
__device__ __forceinline__ void st16(void* base, unsigned off, v4u x) {
  if (ST == 0) asm volatile("global_store_dwordx4 %0, %1, %2 sc1\n\ts_nop 1" ::"v"(off), "v"(x), "s"(base) : "memory");
  if (ST == 1) asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" ::"v"(off), "v"(x), "s"(base) : "memory");
  if (ST == 2) asm volatile("global_store_dwordx4 %0, %1, %2 sc0\n\ts_nop 1" ::"v"(off), "v"(x), "s"(base) : "memory");
  if (ST == 3) asm volatile("global_store_dwordx4 %0, %1, %2 sc0 sc1\n\ts_nop 1" ::"v"(off), "v"(x), "s"(base) : "memory");
}

__global__ void k_xcc(int* out) {
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
  unsigned hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = (int)id; out[2 * blockIdx.x + 1] = (int)hw; }
}

// workgroups a and b play ping-pong: a stores round r into place A, b waits for it and stores r into place B, a waits for it.
// One 16-byte chunk per lane of wave 0 (64 lanes = 1 KB per place, like 32 blocks' half records).  Result: cycles of the 100 MHz wall clock.
template <int LD, int ST>
__global__ __launch_bounds__(256) void k_pingpong(unsigned* buf, int a, int b, int rounds, int limit, long long* out) {
  const int w = blockIdx.x;
  if ((w != a && w != b) || threadIdx.x >= 64) return;
  const bool first = w == a;
  unsigned* mine = buf + (first ? 0 : 4096);
  unsigned* theirs = buf + (first ? 4096 : 0);
  const unsigned off = threadIdx.x * 16;
  long long t0 = wall_clock64();
  int failed = 0;
  for (int r = 1; r <= rounds && !failed; ++r) {
    v4u x = {(unsigned)r, (unsigned)r, (unsigned)r, (unsigned)r};
    if (first) st16<ST>(mine, off, x);
    int polls = 0;
    for (;;) {
      v4u y = ld16<LD>(theirs, off);
      if (__all(y.x == (unsigned)r && y.w == (unsigned)r)) break;
      if (++polls > limit) { failed = r; break; }
    }
    if (!first && !failed) st16<ST>(mine, off, x);
  }
  long long t1 = wall_clock64();
  if (threadIdx.x == 0) { out[first ? 0 : 2] = t1 - t0; out[first ? 1 : 3] = failed; }
}

template <int LD, int ST>
static void run(const char* name, unsigned* buf, long long* out, int a, int b, int rounds, int grid) {
  CK(hipMemset(buf, 0, 8192 * sizeof(unsigned)));
  CK(hipMemset(out, 0, 4 * sizeof(long long)));
  hipLaunchKernelGGL((k_pingpong<LD, ST>), dim3(grid), dim3(256), 0, 0, buf, a, b, rounds, 200000, out);
  CK(hipDeviceSynchronize());
  long long h[4];
  CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
  printf("  %-34s workgroups %3d <-> %3d: %7.1f ns per hand-off%s\n", name, a, b, 10.0 * (double)h[0] / rounds / 2.0,
         (h[1] || h[3]) ? "   (a spin gave up: NOT coherent at this scope / placement)" : "");
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 2000;
  const int grid = 256;
  int* d_x;
  CK(hipMalloc(&d_x, 2 * grid * sizeof(int)));
  hipLaunchKernelGGL(k_xcc, dim3(grid), dim3(256), 0, 0, d_x);
  std::vector<int> x(2 * grid);
  CK(hipMemcpy(x.data(), d_x, x.size() * sizeof(int), hipMemcpyDeviceToHost));
  printf("XCC_ID of workgroups 0..31 of a 256-workgroup launch:");
  for (int w = 0; w < 32; ++w) printf(" %d", x[2 * w] & 15);
  int same = 0;
  for (int w = 0; w < grid; ++w) same += ((x[2 * w] & 15) == (x[2 * (w % 8)] & 15));
  printf("\nworkgroup w on the XCD of workgroup w %% 8: %d of %d\n", same, grid);
  unsigned* buf; long long* out;
  CK(hipMalloc(&buf, 8192 * sizeof(unsigned)));
  CK(hipMalloc(&out, 4 * sizeof(long long)));
  for (int pass = 0; pass < 2; ++pass) {
    printf("pass %d\n", pass);
    const int pairs[3][2] = {{0, 8}, {0, 1}, {0, 128}};
    for (auto& p : pairs) {
      run<0, 0>("store sc1, load sc1 (product)", buf, out, p[0], p[1], rounds, grid);
      run<0, 1>("store plain, load sc1", buf, out, p[0], p[1], rounds, grid);
      run<1, 1>("store plain, load sc0", buf, out, p[0], p[1], rounds, grid);
      run<1, 2>("store sc0, load sc0", buf, out, p[0], p[1], rounds, grid);
      run<1, 0>("store sc1, load sc0", buf, out, p[0], p[1], rounds, grid);
      run<2, 3>("store sc0 sc1, load sc0 sc1", buf, out, p[0], p[1], rounds, grid);
    }
  }
  return 0;
}
