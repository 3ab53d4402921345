// dfx_comm.hip -- the one collective of the path, native: RCCL over xGMI behind three C entry points.
//
// Reference: DifFlexMM's only multi-device construct is a pmap over independent forward inputs
// (problems/quads_kinetic_energy_static_tuning.py:454-478) followed by a host-side sum; here one process per GPU
// integrates its own members with no data-path communication and the per-member objectives (8 B each) are
// all-gathered once per evaluation; gradients w.r.t. a design shared by all ranks (multi-input problems) are summed with
// one all-reduce.  Payloads are tiny (<= a few MB), so the calls are latency-bound: plain ncclAllGather / ncclAllReduce on
// a private stream, host buffers staged through a pinned area.  The unique id is created by rank 0
// (dfx_comm_unique_id) and handed to the other ranks by the launcher (a file or an environment variable).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <string>

#include "../../include/dfx.h"

static thread_local std::string g_comm_error;

struct dfx_comm {
  ncclComm_t comm = nullptr;
  hipStream_t stream = nullptr;
  int rank = 0, nranks = 1, device = 0;
  double* d_buf = nullptr;      // device staging: [send | recv]
  double* h_buf = nullptr;      // pinned host staging
  size_t cap = 0;               // doubles in each of d_buf, h_buf
};

#define COMM_HIP(call)                                                                \
  do {                                                                                \
    hipError_t e_ = (call);                                                           \
    if (e_ != hipSuccess) { g_comm_error = std::string(#call) + ": " + hipGetErrorString(e_); return 2; } \
  } while (0)
#define COMM_NCCL(call)                                                               \
  do {                                                                                \
    ncclResult_t r_ = (call);                                                         \
    if (r_ != ncclSuccess) { g_comm_error = std::string(#call) + ": " + ncclGetErrorString(r_); return 3; } \
  } while (0)

static int ensure(dfx_comm* c, size_t n) {
  if (n <= c->cap) return 0;
  if (c->d_buf) (void)hipFree(c->d_buf);
  if (c->h_buf) (void)hipHostFree(c->h_buf);
  c->d_buf = nullptr; c->h_buf = nullptr; c->cap = 0;
  COMM_HIP(hipMalloc((void**)&c->d_buf, n * sizeof(double)));
  COMM_HIP(hipHostMalloc((void**)&c->h_buf, n * sizeof(double), hipHostMallocDefault));
  c->cap = n;
  return 0;
}

extern "C" {

const char* dfx_comm_last_error(void) { return g_comm_error.c_str(); }

int dfx_comm_unique_id(char* uid128) {
  ncclUniqueId id;
  COMM_NCCL(ncclGetUniqueId(&id));
  static_assert(sizeof(id) == DFX_COMM_UID_BYTES, "unique id size");
  memcpy(uid128, &id, sizeof(id));
  return 0;
}

int dfx_comm_init(int32_t rank, int32_t nranks, const char* uid128, int32_t device, dfx_comm** out) {
  if (!out || !uid128 || nranks < 1 || rank < 0 || rank >= nranks) { g_comm_error = "comm_init: invalid arguments"; return 1; }
  dfx_comm* c = new dfx_comm();
  c->rank = rank; c->nranks = nranks; c->device = device;
  auto fail = [&](int rc) { delete c; return rc; };
  if (hipSetDevice(device) != hipSuccess) { g_comm_error = "comm_init: hipSetDevice failed"; return fail(2); }
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { g_comm_error = "comm_init: hipStreamCreate failed"; return fail(2); }
  { int rt = 0;      // the headers this file was compiled with and the library it runs must agree on the major version
    if (ncclGetVersion(&rt) != ncclSuccess || rt / 10000 != NCCL_VERSION_CODE / 10000) {
      g_comm_error = "comm_init: RCCL runtime " + std::to_string(rt) + " does not match the headers (" + std::to_string(NCCL_VERSION_CODE) + ")";
      (void)hipStreamDestroy(c->stream); return fail(3);
    } }
  ncclUniqueId id;
  memcpy(&id, uid128, sizeof(id));
  ncclResult_t r = ncclCommInitRank(&c->comm, nranks, id, rank);
  if (r != ncclSuccess) { g_comm_error = std::string("ncclCommInitRank: ") + ncclGetErrorString(r); (void)hipStreamDestroy(c->stream); return fail(3); }
  *out = c;
  return 0;
}

int dfx_comm_destroy(dfx_comm* c) {
  if (!c) return 0;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->comm) (void)ncclCommDestroy(c->comm);
  if (c->d_buf) (void)hipFree(c->d_buf);
  if (c->h_buf) (void)hipHostFree(c->h_buf);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return 0;
}

// RCCL the process actually runs (another library with the same SONAME may have been loaded first, e.g. the one bundled with a
// Python package) and the one the engine was compiled against: major * 10000 + minor * 100 + patch
int dfx_comm_rccl_version(int32_t* runtime, int32_t* compiled) {
  int v = 0;
  COMM_NCCL(ncclGetVersion(&v));
  if (runtime) *runtime = v;
  if (compiled) *compiled = NCCL_VERSION_CODE;
  return 0;
}

int dfx_comm_rank(const dfx_comm* c) { return c ? c->rank : 0; }
int dfx_comm_size(const dfx_comm* c) { return c ? c->nranks : 1; }

// all[r * n_local + i] = local_i of rank r, on every rank
int dfx_gather_objectives(dfx_comm* c, const double* local, int32_t n_local, double* all) {
  if (!c || n_local < 0) { g_comm_error = "gather_objectives: invalid arguments"; return 1; }
  if (n_local == 0) return 0;
  COMM_HIP(hipSetDevice(c->device));
  const size_t n = (size_t)n_local, total = n * c->nranks;
  if (int rc = ensure(c, n + total)) return rc;
  memcpy(c->h_buf, local, n * sizeof(double));
  COMM_HIP(hipMemcpyAsync(c->d_buf, c->h_buf, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
  COMM_NCCL(ncclAllGather(c->d_buf, c->d_buf + n, n, ncclDouble, c->comm, c->stream));
  COMM_HIP(hipMemcpyAsync(c->h_buf + n, c->d_buf + n, total * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  COMM_HIP(hipStreamSynchronize(c->stream));
  memcpy(all, c->h_buf + n, total * sizeof(double));
  return 0;
}

// in place, op: DFX_REDUCE_SUM / _MAX / _MIN
int dfx_comm_allreduce(dfx_comm* c, double* inout, int64_t n, int32_t op) {
  if (!c || n < 0) { g_comm_error = "allreduce: invalid arguments"; return 1; }
  if (n == 0) return 0;
  const ncclRedOp_t rop = op == DFX_REDUCE_MAX ? ncclMax : (op == DFX_REDUCE_MIN ? ncclMin : ncclSum);
  COMM_HIP(hipSetDevice(c->device));
  if (int rc = ensure(c, (size_t)n)) return rc;
  memcpy(c->h_buf, inout, (size_t)n * sizeof(double));
  COMM_HIP(hipMemcpyAsync(c->d_buf, c->h_buf, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream));
  COMM_NCCL(ncclAllReduce(c->d_buf, c->d_buf, (size_t)n, ncclDouble, rop, c->comm, c->stream));
  COMM_HIP(hipMemcpyAsync(c->h_buf, c->d_buf, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  COMM_HIP(hipStreamSynchronize(c->stream));
  memcpy(inout, c->h_buf, (size_t)n * sizeof(double));
  return 0;
}

// gradients of a design shared by all ranks: summed in place (problems/quads_focusing_multi_input.py:66-86 sums its inputs)
int dfx_reduce_grads(dfx_comm* c, double* inout, int64_t n) { return dfx_comm_allreduce(c, inout, n, DFX_REDUCE_SUM); }

int dfx_comm_barrier(dfx_comm* c) {
  double one = 1.0;
  return dfx_comm_allreduce(c, &one, 1, DFX_REDUCE_SUM);
}

// ---- device helpers the benchmark needs without any other GPU library
int dfx_mem_info(int32_t device, int64_t* free_bytes, int64_t* total_bytes) {
  size_t f = 0, t = 0;
  if (hipSetDevice(device) != hipSuccess || hipMemGetInfo(&f, &t) != hipSuccess) { g_comm_error = "mem_info: no such HIP device"; return 2; }
  if (free_bytes) *free_bytes = (int64_t)f;
  if (total_bytes) *total_bytes = (int64_t)t;
  return 0;
}

int dfx_device_synchronize(int32_t device) {
  if (hipSetDevice(device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { g_comm_error = "device_synchronize failed"; return 2; }
  return 0;
}

}  // extern "C"
