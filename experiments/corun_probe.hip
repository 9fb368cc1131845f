// corun_probe.hip — can an HBM-streaming kernel and an MFMA-bound kernel share the CUs of an MI355X?
//
// The engine's two LM stream groups are meant to overlap one group's attention (HBM-bound) with the other's GEMMs
// (MFMA-bound).  Measured at B = 2048 they mostly take turns instead (attention 2.45x longer live than alone, step 73 ms
// against 63 + 18 serial).  Suspect: the register file.  The large GEMM kernels hold 2 waves x 168..200 VGPRs per SIMD, which
// leaves room for ONE 128-VGPR attention wave per SIMD (one workgroup per CU) — and a register-staged streaming kernel
// with 4 waves per CU cannot keep enough bytes in flight to load HBM.  This probe prices the alternatives in isolation:
//
//   stream_reg<UNR>  register-staged streaming, two sets of UNR x 16 B per lane in flight (the attention kernel's scheme)
//   stream_dma<R>    LDS-DMA streaming (global_load_lds_dwordx4): wave-private ring of R x 1 KiB slots in LDS, the bytes in
//                    flight cost no VGPRs; each lane reads its own 16 B back with ds_read_b128 after a counted vmcnt
//   mfma_burn<VG>    v_mfma_f32_16x16x4_f32 on 8 accumulators, VGPR allocation forced to VG, 2 workgroups per CU
//
// alone, and the streaming kernels beside a resident mfma_burn (separate streams).  Every streaming run checks the sum of
// all dwords it consumed.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP %s @%d\n", hipGetErrorString(e_), __LINE__); exit(2);} } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __host__ inline uint32_t pattern(uint64_t i) { return (uint32_t)(i * 2654435761ull) ^ (uint32_t)(i >> 13); }

__device__ unsigned long long g_ts[4];  // [0] stream first-in, [1] stream last-out, [2] burn first-in, [3] burn last-out (100 MHz)
__device__ __forceinline__ void stamp_in(int k) { if (threadIdx.x == 0) atomicMin(&g_ts[k], (unsigned long long)wall_clock64()); }
__device__ __forceinline__ void stamp_out(int k) { if (threadIdx.x == 0) atomicMax(&g_ts[k], (unsigned long long)wall_clock64()); }

__global__ void fill(uint32_t* p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = pattern(i);
}

// one wave-instruction: 64 lanes x 16 B from per-lane global addresses to LDS [lds_dst, lds_dst + 1024)
__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

template <int R, int PRIO = 0>
__global__ __launch_bounds__(256) void stream_dma(const char* __restrict__ src, int items_per_wave, uint32_t* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  if (PRIO) __builtin_amdgcn_s_setprio(3);
  stamp_in(0);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const char* base = src + (size_t)blockIdx.x * items_per_wave * 4096 + lane * 16;
  const uint32_t ring = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds + w * R * 1024);
  const int last = items_per_wave - 1;
#pragma unroll
  for (int i = 0; i < R; ++i) glds16(base + (size_t)(min(i, last) * 4 + w) * 1024, ring + i * 1024);
  uint32_t sum = 0;
  int slot = 0;
  for (int i = 0; i < items_per_wave; ++i) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(R - 1) : "memory");  // R issued beyond item i-1: at most R-1 outstanding = item i landed
    const uint4 v = *reinterpret_cast<const uint4*>(lds + (w * R + slot) * 1024 + lane * 16);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the slot is in registers before it is refilled
    glds16(base + (size_t)(min(i + R, last) * 4 + w) * 1024, ring + slot * 1024);
    sum += v.x + v.y + v.z + v.w;
    slot = slot + 1 == R ? 0 : slot + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off, 64);
  if (lane == 0) atomicAdd(out, sum);
  stamp_out(1);
}

template <int UNR, int PRIO = 0>
__global__ __launch_bounds__(256, 4) void stream_reg(const char* __restrict__ src, int items_per_wave, uint32_t* __restrict__ out) {
  if (PRIO) __builtin_amdgcn_s_setprio(3);
  stamp_in(0);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const char* base = src + (size_t)blockIdx.x * items_per_wave * 4096 + lane * 16;
  const int last = items_per_wave - 1;
  uint4 ra[UNR], rb[UNR];
  asm volatile("; claim the attention kernel's allocation" ::: "v127");
#define ISSUE(Rg, I0) _Pragma("unroll") for (int u = 0; u < UNR; ++u) Rg[u] = *reinterpret_cast<const uint4*>(base + (size_t)(min((I0) + u, last) * 4 + w) * 1024);
#define USE(Rg, I0) _Pragma("unroll") for (int u = 0; u < UNR; ++u) if ((I0) + u <= last) sum += Rg[u].x + Rg[u].y + Rg[u].z + Rg[u].w;
  uint32_t sum = 0;
  int i = 0;
  ISSUE(ra, 0)
  while (i < items_per_wave) {
    ISSUE(rb, i + UNR)
    USE(ra, i)
    i += UNR;
    if (i >= items_per_wave) break;
    ISSUE(ra, i + UNR)
    USE(rb, i)
    i += UNR;
  }
#undef ISSUE
#undef USE
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off, 64);
  if (lane == 0) atomicAdd(out, sum);
  stamp_out(1);
}

template <int VG>
__global__ __launch_bounds__(256, 2) void mfma_burn(float* out, int iters, float a0, float b0) {
  stamp_in(2);
  f32x4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 1e-3f;
  if (VG == 192) asm volatile("; claim" ::: "v191");
  if (VG == 168) asm volatile("; claim" ::: "v167");
  if (VG == 200) asm volatile("; claim" ::: "v199");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678f) out[0] = s;
  stamp_out(3);
}

struct Ctx {
  char* src;
  size_t bytes;
  int wgs, items_per_wave;
  uint32_t* out;
  uint32_t want;
  float* mf;
  hipStream_t sa, sb;
};

template <typename F>
double time_ms(F&& f, int reps = 3) {
  double best = 1e30;
  for (int r = 0; r < reps; ++r) {
    CK(hipDeviceSynchronize());
    auto t0 = std::chrono::steady_clock::now();
    f();
    CK(hipDeviceSynchronize());
    best = std::min(best, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  }
  return best;
}

template <typename K>
void stream_case(Ctx& c, const char* name, K launch, size_t lds, int burn_iters, double t_burn_alone, int burn_wgs = 512, int vg = 192) {
  auto run_a = [&] { CK(hipMemsetAsync(c.out, 0, 4, c.sa)); launch(c.sa, lds); };
  const double ta = time_ms(run_a);
  uint32_t got = 0;
  CK(hipMemcpy(&got, c.out, 4, hipMemcpyDeviceToHost));
  printf("%-34s alone %7.3f ms  %6.0f GB/s  %s", name, ta, c.bytes / ta / 1e6, got == c.want ? "sum ok" : "SUM WRONG");
  if (burn_iters > 0) {
    hipEvent_t a0, a1, b0, b1;
    CK(hipEventCreate(&a0)); CK(hipEventCreate(&a1)); CK(hipEventCreate(&b0)); CK(hipEventCreate(&b1));
    double tab = 1e30;
    float da = 0, db = 0;
    unsigned long long ts[4] = {0, 0, 0, 0};
    for (int r = 0; r < 3; ++r) {
      const unsigned long long init[4] = {~0ull, 0ull, ~0ull, 0ull};
      CK(hipMemcpyToSymbol(HIP_SYMBOL(g_ts), init, sizeof init));
      CK(hipDeviceSynchronize());
      auto t0 = std::chrono::steady_clock::now();
      CK(hipEventRecord(b0, c.sb));
      const int bi = (int)((long)burn_iters * 512 / burn_wgs);
      if (vg == 128) hipLaunchKernelGGL(mfma_burn<128>, dim3(burn_wgs), dim3(256), 0, c.sb, c.mf, bi, 1.2345f, 0.9876f);
      else if (vg == 168) hipLaunchKernelGGL(mfma_burn<168>, dim3(burn_wgs), dim3(256), 0, c.sb, c.mf, bi, 1.2345f, 0.9876f);
      else if (vg == 200) hipLaunchKernelGGL(mfma_burn<200>, dim3(burn_wgs), dim3(256), 0, c.sb, c.mf, bi, 1.2345f, 0.9876f);
      else hipLaunchKernelGGL(mfma_burn<192>, dim3(burn_wgs), dim3(256), 0, c.sb, c.mf, bi, 1.2345f, 0.9876f);
      CK(hipEventRecord(b1, c.sb));
      CK(hipMemsetAsync(c.out, 0, 4, c.sa));
      CK(hipEventRecord(a0, c.sa));
      launch(c.sa, lds);
      CK(hipEventRecord(a1, c.sa));
      CK(hipDeviceSynchronize());
      const double t = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      if (t < tab) {
        tab = t; CK(hipEventElapsedTime(&da, a0, a1)); CK(hipEventElapsedTime(&db, b0, b1));
        CK(hipMemcpyFromSymbol(ts, HIP_SYMBOL(g_ts), sizeof ts));
      }
    }
    CK(hipMemcpy(&got, c.out, 4, hipMemcpyDeviceToHost));
    printf(" | beside mfma_burn<%d> x%d (alone %.3f): both %7.3f ms (stream %.3f = %4.0f GB/s, burn %.3f; serial %.3f, ideal %.3f) %s",
           vg, burn_wgs, t_burn_alone, tab, da, c.bytes / da / 1e6, db, ta + t_burn_alone, std::max(ta, t_burn_alone), got == c.want ? "ok" : "SUM WRONG");
    const double o = (double)std::min(ts[0], ts[2]);
    printf("\n%38s device clock, us from the first workgroup: stream [%.0f, %.0f]  burn x%d [%.0f, %.0f]", "", (ts[0] - o) / 100.0, (ts[1] - o) / 100.0,
           burn_wgs, (ts[2] - o) / 100.0, (ts[3] - o) / 100.0);
  }
  printf("\n");
  fflush(stdout);
}

int main(int argc, char** argv) {
  Ctx c;
  c.wgs = argc > 1 ? atoi(argv[1]) : 8192;
  c.items_per_wave = 96;  // 4 waves x 96 KiB = 384 KiB per workgroup: K + V of one (slot, head) of stt-1b at 750 frames
  c.bytes = (size_t)c.wgs * c.items_per_wave * 4096;
  CK(hipMalloc(&c.src, c.bytes));
  CK(hipMalloc(&c.out, 4));
  CK(hipMalloc(&c.mf, 4096));
  CK(hipStreamCreateWithFlags(&c.sa, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&c.sb, hipStreamNonBlocking));
  hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, (uint32_t*)c.src, c.bytes / 4);
  CK(hipDeviceSynchronize());
  {
    uint32_t s = 0;
    for (uint64_t i = 0; i < c.bytes / 4; ++i) s += pattern(i);
    c.want = s;
  }
  printf("streaming %d workgroups x %d KiB = %.2f GB\n", c.wgs, c.items_per_wave * 4, c.bytes / 1e9);

  // size the MFMA kernel to about the streaming time alone
  int burn_iters = 4000;
  double tb = time_ms([&] { hipLaunchKernelGGL(mfma_burn<192>, dim3(512), dim3(256), 0, c.sb, c.mf, burn_iters, 1.2345f, 0.9876f); });
  const double target = c.bytes / 6.0e9;  // ms at 6 TB/s
  burn_iters = (int)(burn_iters * target / tb);
  tb = time_ms([&] { hipLaunchKernelGGL(mfma_burn<192>, dim3(512), dim3(256), 0, c.sb, c.mf, burn_iters, 1.2345f, 0.9876f); });
  printf("mfma_burn<192> x512 workgroups, %d iterations: %.3f ms alone = %.1f TFLOP/s\n", burn_iters, tb,
         (double)burn_iters * 16 * 8 * 512 * 4 * 2048 / (tb * 1e-3) / 1e12);
  for (int vg : {128, 168, 200}) {
    double t = time_ms([&] {
      if (vg == 128) hipLaunchKernelGGL(mfma_burn<128>, dim3(512), dim3(256), 0, c.sb, c.mf, burn_iters, 1.2345f, 0.9876f);
      if (vg == 168) hipLaunchKernelGGL(mfma_burn<168>, dim3(512), dim3(256), 0, c.sb, c.mf, burn_iters, 1.2345f, 0.9876f);
      if (vg == 200) hipLaunchKernelGGL(mfma_burn<200>, dim3(512), dim3(256), 0, c.sb, c.mf, burn_iters, 1.2345f, 0.9876f);
    });
    printf("mfma_burn<%d>: %.3f ms\n", vg, t);
  }

#define DMA(Rv) [&](hipStream_t st, size_t lds) { \
    static bool attr = false; \
    if (!attr) { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&stream_dma<Rv>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr = true; } \
    hipLaunchKernelGGL(stream_dma<Rv>, dim3(c.wgs), dim3(256), lds, st, c.src, c.items_per_wave, c.out); }
#define REG(Uv) [&](hipStream_t st, size_t lds) { \
    static bool attr = false; \
    if (!attr) { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&stream_reg<Uv>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr = true; } \
    hipLaunchKernelGGL(stream_reg<Uv>, dim3(c.wgs), dim3(256), lds, st, c.src, c.items_per_wave, c.out); }
#define DMAP(Rv) [&](hipStream_t st, size_t lds) { \
    static bool attr = false; \
    if (!attr) { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&stream_dma<Rv, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr = true; } \
    hipLaunchKernelGGL((stream_dma<Rv, 1>), dim3(c.wgs), dim3(256), lds, st, c.src, c.items_per_wave, c.out); }
#define REGP(Uv) [&](hipStream_t st, size_t lds) { \
    static bool attr = false; \
    if (!attr) { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&stream_reg<Uv, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr = true; } \
    hipLaunchKernelGGL((stream_reg<Uv, 1>), dim3(c.wgs), dim3(256), lds, st, c.src, c.items_per_wave, c.out); }
  // LDS claims decide the workgroups per CU: 160 KiB / claim
  stream_case(c, "reg UNR=8, 4 wg/CU", REG(8), 36 * 1024, burn_iters, tb);
  stream_case(c, "reg UNR=8, 2 wg/CU", REG(8), 60 * 1024, burn_iters, tb);
  stream_case(c, "reg UNR=8, 1 wg/CU", REG(8), 100 * 1024, burn_iters, tb);
  stream_case(c, "dma R=8  (32 KiB), 4 wg/CU", DMA(8), 36 * 1024, burn_iters, tb);
  stream_case(c, "dma R=15 (60 KiB), 2 wg/CU", DMA(15), 60 * 1024, burn_iters, tb);
  stream_case(c, "dma R=15 (60 KiB), 1 wg/CU", DMA(15), 100 * 1024, burn_iters, tb);
  stream_case(c, "dma R=24 (96 KiB), 1 wg/CU", DMA(24), 100 * 1024, burn_iters, tb);
  stream_case(c, "dma R=30 (120 KiB), 1 wg/CU", DMA(30), 124 * 1024, burn_iters, tb);
  printf("-- what blocks the streaming kernel beside RESIDENT MFMA workgroups?  (mfma_burn<128> really allocates 34 VGPRs)\n");
  stream_case(c, "dma R=8, burn 34 VGPR x512", DMA(8), 36 * 1024, burn_iters, tb, 512, 128);
  stream_case(c, "dma R=8, burn 168 VGPR x512", DMA(8), 36 * 1024, burn_iters, tb, 512, 168);
  stream_case(c, "dma R=8, burn 192 VGPR x256", DMA(8), 36 * 1024, burn_iters, tb, 256, 192);
  stream_case(c, "dma R=8, burn 34 VGPR x256", DMA(8), 36 * 1024, burn_iters, tb, 256, 128);
  stream_case(c, "dma R=8, burn 34 VGPR x1024", DMA(8), 36 * 1024, burn_iters, tb, 1024, 128);
  printf("-- the MFMA kernel as 8192 short workgroups (a GEMM's shape: slots keep turning over)\n");
  stream_case(c, "reg UNR=8, 4 wg/CU", REG(8), 36 * 1024, burn_iters, tb, 8192);
  stream_case(c, "reg UNR=8, 2 wg/CU", REG(8), 60 * 1024, burn_iters, tb, 8192);
  stream_case(c, "reg UNR=8, 1 wg/CU", REG(8), 100 * 1024, burn_iters, tb, 8192);
  stream_case(c, "dma R=15 (60 KiB), 2 wg/CU", DMA(15), 60 * 1024, burn_iters, tb, 8192);
  stream_case(c, "dma R=24 (96 KiB), 1 wg/CU", DMA(24), 100 * 1024, burn_iters, tb, 8192);
  printf("-- the same with s_setprio 3 in the streaming waves\n");
  stream_case(c, "reg UNR=8 prio3, 4 wg/CU", REGP(8), 36 * 1024, burn_iters, tb, 8192);
  stream_case(c, "reg UNR=8 prio3, 2 wg/CU", REGP(8), 60 * 1024, burn_iters, tb, 8192);
  stream_case(c, "reg UNR=8 prio3, 1 wg/CU", REGP(8), 100 * 1024, burn_iters, tb, 8192);
  stream_case(c, "dma R=15 prio3, 2 wg/CU", DMAP(15), 60 * 1024, burn_iters, tb, 8192);
  stream_case(c, "dma R=24 prio3, 1 wg/CU", DMAP(24), 100 * 1024, burn_iters, tb, 8192);
  stream_case(c, "dma R=24 prio3, 1 wg/CU, resident burn", DMAP(24), 100 * 1024, burn_iters, tb, 512);
  return 0;
}
