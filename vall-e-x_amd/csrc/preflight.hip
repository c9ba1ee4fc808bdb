// Box pre-flight (vx_preflight.bin): a LIBRARY-FREE check that this lease can do what every call of libvallex_hip.so needs before any
// of its kernels runs -- enumerate the GPU, allocate, move 8 MiB host -> device -> host and run a kernel.  It is what tells a
// lease-level fault from a product fault in the driver's record (round 5: the suite died inside the first weight upload,
// vx_load_tensor, replacing load_state_dict + .to(device) of utils/generation.py:79-83, with nothing to say which it was).
// Two modes, run as separate processes by tests/test_gpu_a0_preflight.py and __graft_entry__.smoke() (a GPU memory fault aborts
// the process that caused it; it must not be pytest):
//   vx_preflight.bin pinned     the product's transfer path: hipHostMalloc ring + hipMemcpyAsync on a non-blocking stream
//   vx_preflight.bin pageable   plain hipMemcpy of malloc'd memory (the runtime pins user pages on the fly) -- informational:
//                               the library no longer depends on it
// Prints one JSON line; exit 0 = ok, 1 = wrong data / HIP error (a fault kills it with SIGABRT: the caller reports the signal).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

__global__ void add_one(unsigned* p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] += 1u;
}

#define CK(x)                                                                                  \
  do {                                                                                         \
    hipError_t e_ = (x);                                                                       \
    if (e_ != hipSuccess) {                                                                    \
      printf("{\"ok\": false, \"mode\": \"%s\", \"error\": \"%s: %s\"}\n", mode, #x, hipGetErrorString(e_)); \
      return 1;                                                                                \
    }                                                                                          \
  } while (0)

int main(int argc, char** argv) {
  const char* mode = argc > 1 ? argv[1] : "pinned";
  const bool pinned = strcmp(mode, "pageable") != 0;
  const size_t n = (8u << 20) / sizeof(unsigned);
  int ndev = 0, drv = 0, rt = 0;
  CK(hipGetDeviceCount(&ndev));
  if (ndev <= 0) { printf("{\"ok\": false, \"mode\": \"%s\", \"error\": \"no HIP device\"}\n", mode); return 1; }
  CK(hipSetDevice(0));
  hipDeviceProp_t pr;
  CK(hipGetDeviceProperties(&pr, 0));
  (void)hipDriverGetVersion(&drv);
  (void)hipRuntimeGetVersion(&rt);
  unsigned *h = nullptr, *d = nullptr;
  if (pinned) CK(hipHostMalloc((void**)&h, n * sizeof(unsigned), hipHostMallocDefault));
  else h = (unsigned*)malloc(n * sizeof(unsigned));
  if (!h) { printf("{\"ok\": false, \"mode\": \"%s\", \"error\": \"host allocation\"}\n", mode); return 1; }
  for (size_t i = 0; i < n; ++i) h[i] = (unsigned)(i * 2654435761u);
  CK(hipMalloc((void**)&d, n * sizeof(unsigned)));
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  if (pinned) CK(hipMemcpyAsync(d, h, n * sizeof(unsigned), hipMemcpyHostToDevice, st));
  else { CK(hipMemcpy(d, h, n * sizeof(unsigned), hipMemcpyHostToDevice)); }
  add_one<<<256, 256, 0, st>>>(d, n);
  CK(hipGetLastError());
  CK(hipStreamSynchronize(st));
  memset(h, 0, n * sizeof(unsigned));
  if (pinned) { CK(hipMemcpyAsync(h, d, n * sizeof(unsigned), hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); }
  else CK(hipMemcpy(h, d, n * sizeof(unsigned), hipMemcpyDeviceToHost));
  size_t bad = 0;
  for (size_t i = 0; i < n; ++i) bad += h[i] != (unsigned)(i * 2654435761u) + 1u;
  printf("{\"ok\": %s, \"mode\": \"%s\", \"bytes\": %zu, \"mismatches\": %zu, \"devices\": %d, \"name\": \"%s\", \"arch\": \"%s\", "
         "\"cus\": %d, \"hbm_gib\": %.1f, \"driver\": %d, \"runtime\": %d}\n",
         bad ? "false" : "true", mode, n * sizeof(unsigned), bad, ndev, pr.name, pr.gcnArchName, pr.multiProcessorCount,
         (double)pr.totalGlobalMem / (1 << 30), drv, rt);
  return bad ? 1 : 0;
}
