// How long does device memory take to ALLOCATE on this platform, and can it be had faster?  (the records checkpoint of one C3 design over
// its whole horizon is 275 GB: hipMalloc took 4.5 s of a 6 s first solve)  hipcc --offload-arch=gfx950 -O2 -o alloc_probe alloc_probe.hip -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void touch(double* p, size_t n, size_t stride) { size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * stride; if (i < n) p[i] = 1.0; }
int main(int argc, char** argv) {
  const size_t GB = 1ull << 30;
  const size_t total = (argc > 1 ? atoll(argv[1]) : 32) * GB;
  OK(hipSetDevice(0));
  OK(hipFree(nullptr));
  { // 1. one hipMalloc
    double t0 = now(); void* p = nullptr; OK(hipMalloc(&p, total)); double t1 = now();
    hipLaunchKernelGGL(touch, dim3((unsigned)(total / 8 / 512 / 256 + 1)), dim3(256), 0, 0, (double*)p, total / 8, (size_t)512); OK(hipDeviceSynchronize()); double t2 = now();
    OK(hipFree(p)); double t3 = now();
    printf("hipMalloc %zu GB: %.3f s (%.1f ms/GB), first touch of every page %.3f s, hipFree %.3f s\n", total / GB, t1 - t0, 1e3 * (t1 - t0) / (total / GB), t2 - t1, t3 - t2);
  }
  { // 2. again (does the driver keep anything?)
    double t0 = now(); void* p = nullptr; OK(hipMalloc(&p, total)); double t1 = now(); OK(hipFree(p));
    printf("hipMalloc %zu GB again: %.3f s\n", total / GB, t1 - t0);
  }
  for (int nt : {2, 4, 8}) { // 3. the same bytes as nt allocations from nt host threads
    std::vector<void*> ps(nt, nullptr); std::vector<std::thread> th;
    double t0 = now();
    for (int k = 0; k < nt; ++k) th.emplace_back([&, k] { (void)hipSetDevice(0); (void)hipMalloc(&ps[k], total / nt); });
    for (auto& t : th) t.join();
    double t1 = now();
    for (void* p : ps) OK(hipFree(p));
    printf("%d threads x hipMalloc %zu GB: %.3f s\n", nt, total / nt / GB, t1 - t0);
  }
  { // 4. virtual memory management: one address range, physical chunks created and mapped
    hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gran = 0; OK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    printf("VMM granularity %zu\n", gran);
    for (int nt : {1, 8}) {
      const size_t chunk = 2 * GB, nchunk = total / chunk;
      void* va = nullptr; double t0 = now(); OK(hipMemAddressReserve(&va, total, gran, nullptr, 0)); double t1 = now();
      std::vector<hipMemGenericAllocationHandle_t> hs(nchunk);
      std::vector<std::thread> th;
      for (int k = 0; k < nt; ++k) th.emplace_back([&, k] {
        (void)hipSetDevice(0);
        for (size_t c = k; c < nchunk; c += nt) {
          OK(hipMemCreate(&hs[c], chunk, &prop, 0));
          OK(hipMemMap((char*)va + c * chunk, chunk, 0, hs[c], 0));
          hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
          OK(hipMemSetAccess((char*)va + c * chunk, chunk, &acc, 1));
        }});
      for (auto& t : th) t.join();
      double t2 = now();
      hipLaunchKernelGGL(touch, dim3((unsigned)(total / 8 / 512 / 256 + 1)), dim3(256), 0, 0, (double*)va, total / 8, (size_t)512); OK(hipDeviceSynchronize()); double t3 = now();
      for (size_t c = 0; c < nchunk; ++c) { OK(hipMemUnmap((char*)va + c * chunk, chunk)); OK(hipMemRelease(hs[c])); }
      OK(hipMemAddressFree(va, total));
      printf("VMM %zu GB in %zu chunks, %d thread(s): reserve %.4f s, create+map+access %.3f s, touch %.3f s, teardown %.3f s\n", total / GB, nchunk, nt, t1 - t0, t2 - t1, t3 - t2, now() - t3);
    }
  }
  { // 5. hipMallocAsync from the default pool
    void* p = nullptr; double t0 = now(); OK(hipMallocAsync(&p, total, 0)); OK(hipStreamSynchronize(0)); double t1 = now(); OK(hipFreeAsync(p, 0)); OK(hipStreamSynchronize(0));
    printf("hipMallocAsync %zu GB: %.3f s\n", total / GB, t1 - t0);
  }
  return 0;
}
