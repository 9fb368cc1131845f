// queue_probe.hip — do two HIP streams of this process dispatch concurrently at all?  Kernel S sleeps (s_sleep loop on the
// 100 MHz clock, one wave per CU, no resources to speak of); kernel T stamps its start.  T on another stream should start
// while S sleeps.  Variants: stream creation flags / priorities, launch order, an event record in front of T.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP %s @%d\n", hipGetErrorString(e_), __LINE__); exit(2);} } while (0)
__device__ unsigned long long g_ts[4];
__global__ void sleeper(unsigned long long ticks) {
  if (threadIdx.x == 0) atomicMin(&g_ts[0], (unsigned long long)wall_clock64());
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
  if (threadIdx.x == 0) atomicMax(&g_ts[1], (unsigned long long)wall_clock64());
}
__global__ void toucher(int* p) {
  if (threadIdx.x == 0) atomicMin(&g_ts[2], (unsigned long long)wall_clock64());
  if (p) p[blockIdx.x] = 1;
  if (threadIdx.x == 0) atomicMax(&g_ts[3], (unsigned long long)wall_clock64());
}
void run(const char* name, hipStream_t ss, hipStream_t st, int sleeper_wgs, int sleeper_threads, bool event_first, bool memset_first, int* buf) {
  const unsigned long long init[4] = {~0ull, 0ull, ~0ull, 0ull};
  CK(hipMemcpyToSymbol(HIP_SYMBOL(g_ts), init, sizeof init));
  CK(hipDeviceSynchronize());
  hipEvent_t ev; CK(hipEventCreate(&ev));
  hipLaunchKernelGGL(sleeper, dim3(sleeper_wgs), dim3(sleeper_threads), 0, ss, 50000ull);  // 500 us
  if (memset_first) CK(hipMemsetAsync(buf, 0, 4, st));
  if (event_first) CK(hipEventRecord(ev, st));
  hipLaunchKernelGGL(toucher, dim3(1024), dim3(256), 0, st, buf);
  CK(hipDeviceSynchronize());
  unsigned long long ts[4];
  CK(hipMemcpyFromSymbol(ts, HIP_SYMBOL(g_ts), sizeof ts));
  printf("%-70s sleeper [0, %.0f] us, toucher [%.0f, %.0f] us\n", name, (ts[1] - ts[0]) / 100.0, ((double)ts[2] - (double)ts[0]) / 100.0, ((double)ts[3] - (double)ts[0]) / 100.0);
}
int main() {
  int* buf; CK(hipMalloc(&buf, 4096 * 4));
  hipStream_t a, b, c, d, hi, lo;
  CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
  CK(hipStreamCreate(&c));
  CK(hipStreamCreate(&d));
  int plo, phi; CK(hipDeviceGetStreamPriorityRange(&plo, &phi));
  CK(hipStreamCreateWithPriority(&hi, hipStreamNonBlocking, phi));
  CK(hipStreamCreateWithPriority(&lo, hipStreamNonBlocking, plo));
  for (int rep = 0; rep < 2; ++rep) {
    run("nonblocking a -> nonblocking b, 256 x 64 sleeper", a, b, 256, 64, false, false, buf);
    run("nonblocking a -> nonblocking b, 512 x 256 sleeper", a, b, 512, 256, false, false, buf);
    run("nonblocking a -> nonblocking b, event recorded before toucher", a, b, 256, 64, true, false, buf);
    run("nonblocking a -> nonblocking b, memsetAsync before toucher", a, b, 256, 64, false, true, buf);
    run("default-flag c -> default-flag d", c, d, 256, 64, false, false, buf);
    run("priority hi -> priority lo", hi, lo, 256, 64, false, false, buf);
    run("priority lo -> priority hi", lo, hi, 256, 64, false, false, buf);
    run("priority hi -> nonblocking a", hi, a, 256, 64, false, false, buf);
    run("same stream a -> a (must serialize)", a, a, 256, 64, false, false, buf);
  }
  return 0;
}
