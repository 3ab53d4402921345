#include <hip/hip_runtime.h>
__device__ __forceinline__ double dppd_shl(double v, int) { return v; }
template <int CTRL> __device__ __forceinline__ double dpp_mov64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double swap32(double v) {   // lane i < 32 receives lane i + 32
  unsigned lo = __double2loint(v), hi = __double2hiint(v);
  auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double((int)b[1], (int)a[1]);
}
__device__ __forceinline__ double swap16(double v) {
  unsigned lo = __double2loint(v), hi = __double2hiint(v);
  auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  return __hiloint2double((int)b[1], (int)a[1]);
}
__device__ __forceinline__ double wave_sum_tree(double v) {
  v += swap32(v);
  v += swap16(v);
  v += dpp_mov64<0x108>(v);   // row_shl:8
  v += dpp_mov64<0x104>(v);
  v += dpp_mov64<0x102>(v);
  v += dpp_mov64<0x101>(v);
  return v;
}
__global__ void k(const double* in, double* out) {
  double v = in[threadIdx.x];
  double a = v;
  for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off, 64);
  double b = wave_sum_tree(v);
  if (threadIdx.x == 0) { out[0] = a; out[1] = b; }
  // which element of the swap result is what: dump
  unsigned x = threadIdx.x;
  auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
  auto r16 = __builtin_amdgcn_permlane16_swap(x, x, false, false);
  out[2 + threadIdx.x] = r[0] * 1000.0 + r[1];
  out[70 + threadIdx.x] = r16[0] * 1000.0 + r16[1];
}
int main() {
  double h[64], *d_in, *d_out, o[140];
  for (int i = 0; i < 64; ++i) h[i] = 1.0 / (3.0 + i * 0.7) + 1e-9 * i * i;
  hipMalloc(&d_in, sizeof(h)); hipMalloc(&d_out, sizeof(o));
  hipMemcpy(d_in, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_in, d_out);
  hipMemcpy(o, d_out, sizeof(o), hipMemcpyDeviceToHost);
  printf("shfl tree %.17g  dpp tree %.17g  %s\n", o[0], o[1], o[0] == o[1] ? "IDENTICAL" : "DIFFERENT");
  printf("permlane32_swap lanes 0,1,31,32,33,63: "); for (int i : {0,1,31,32,33,63}) printf("%g ", o[2+i]); printf("\n");
  printf("permlane16_swap lanes 0,1,15,16,17,31,32,47,48,63: "); for (int i : {0,1,15,16,17,31,32,47,48,63}) printf("%g ", o[70+i]); printf("\n");
  return 0;
}
