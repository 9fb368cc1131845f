// mall_probe.hip — what is a weight matrix worth in the Infinity Cache?  A GEMM-sized buffer (8 .. 103 MB) is streamed once by
// every CU (16-byte loads, 8 in flight per lane) (a) cold: right after 1 GiB of another buffer went through the chip,
// (b) warm: again at once, (c) after 1 GiB of NON-TEMPORAL traffic (what the ring caches are since r02).  Device clock,
// HIP events around the launch (a constant launch overhead in every column).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP %s @%d\n", hipGetErrorString(e_), __LINE__); exit(2);} } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ unsigned long long g_ts[2];
template <int NT>
__global__ __launch_bounds__(256) void stream(const u32x4* __restrict__ p, size_t n16, unsigned* out) {
  unsigned s = 0;
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 7 * stride < n16; i += 8 * stride) {
    u32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = NT ? __builtin_nontemporal_load(p + i + u * stride) : p[i + u * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u][0] + v[u][1] + v[u][2] + v[u][3];
  }
  for (; i < n16; i += stride) { const u32x4 v = p[i]; s += v[0] + v[3]; }
  if (s == 0x12345678u) out[0] = s;
}
static double timed(void (*k)(const u32x4*, size_t, unsigned*), const u32x4* p, size_t bytes, unsigned* out, int wgs) {
  static hipEvent_t e0 = nullptr, e1 = nullptr;
  if (!e0) { CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); }
  CK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 0, 0, p, bytes / 16, out);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1000.0;  // us, launch overhead included (the same in every column)
}
int main() {
  const size_t big = 1ull << 30;
  char *w, *other; unsigned* out;
  CK(hipMalloc(&w, 128u << 20)); CK(hipMalloc(&other, big)); CK(hipMalloc(&out, 64));
  CK(hipMemset(w, 1, 128u << 20)); CK(hipMemset(other, 2, big));
  for (size_t mb : {8, 25, 46, 103}) {
    const size_t bytes = mb << 20;
    for (int rep = 0; rep < 2; ++rep) {
      (void)timed(stream<0>, (const u32x4*)other, big, out, 2048);
      const double cold = timed(stream<0>, (const u32x4*)w, bytes, out, 768);
      const double warm = timed(stream<0>, (const u32x4*)w, bytes, out, 768);
      (void)timed(stream<1>, (const u32x4*)other, big, out, 2048);
      const double after_nt = timed(stream<0>, (const u32x4*)w, bytes, out, 768);
      (void)timed(stream<0>, (const u32x4*)other, 200u << 20, out, 2048);
      const double after_200 = timed(stream<0>, (const u32x4*)w, bytes, out, 768);
      printf("%4zu MB: cold %6.1f us (%5.2f TB/s)  warm %6.1f us (%5.2f TB/s)  after 1 GiB of nt loads %6.1f us  after 200 MB of plain loads %6.1f us\n",
             mb, cold, bytes / cold / 1e6, warm, bytes / warm / 1e6, after_nt, after_200);
    }
  }
  return 0;
}
