// grid_barrier_probe.hip — what does a device-wide barrier cost inside one kernel on MI355X? (r03)
// Decides whether the TTS depformer (≈1 300 dependent launches of 4-7 us per step) is worth rebuilding as a persistent kernel
// with grid barriers between its phases.  G workgroups of 256 threads (all co-resident), each phase writes a little data that
// a workgroup of ANOTHER XCD reads in the next phase (so the barrier has to carry device-scope release/acquire), N barriers.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP %s @%d\n", hipGetErrorString(e_), __LINE__); exit(2);} } while (0)

__device__ __forceinline__ void grid_barrier(unsigned* ctr, unsigned& epoch, unsigned G) {
  __syncthreads();
  if (threadIdx.x == 0) {
    epoch += G;
    __threadfence();  // release: this workgroup's stores before the arrival
    atomicAdd(ctr, 1u);
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) __builtin_amdgcn_s_sleep(1);
    __threadfence();  // acquire
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void barrier_kernel(unsigned* ctr, float* buf, int nbar, int work, float* out) {
  const unsigned G = gridDim.x;
  unsigned epoch = 0;
  float acc = 0.f;
  for (int i = 0; i < nbar; ++i) {
    // produce: 1 KB per workgroup; consume: the neighbour's (blockIdx + 1: round-robin dispatch puts it on the next XCD)
    buf[(size_t)blockIdx.x * 256 + threadIdx.x] = acc + (float)i;
    grid_barrier(ctr, epoch, G);
    float v = buf[(size_t)((blockIdx.x + 1) % G) * 256 + threadIdx.x];
    for (int w = 0; w < work; ++w) v = v * 1.0001f + 0.5f;
    acc += v;
  }
  if (acc == 12345.678f) out[0] = acc;
}

int main() {
  unsigned* ctr; float *buf, *out;
  CK(hipMalloc(&ctr, 4)); CK(hipMalloc(&buf, 1024 * 256 * 4)); CK(hipMalloc(&out, 4));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int G : {64, 128, 256, 512}) for (int nbar : {100, 1000}) {
    CK(hipMemset(ctr, 0, 4));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(barrier_kernel, dim3(G), dim3(256), 0, 0, ctr, buf, nbar, 0, out);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("G=%3d workgroups, %4d barriers: %.3f ms total, %.2f us per barrier\n", G, nbar, ms, ms * 1000 / nbar);
  }
  return 0;
}
