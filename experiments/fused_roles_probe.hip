// fused_roles_probe.hip — r03: can HBM streaming and f32-MFMA work share a CU when they are launched as ONE kernel?
//
// r02's corun_probe launched the two kinds of work as separate kernels on separate streams and saw them take turns.  Two
// things were mixed up in that result: what the dispatcher does between queues, and what two kinds of waves do to each other
// once they ARE resident on the same SIMD.  This probe removes the dispatcher: one launch, 512-thread workgroups (one per CU,
// or two), waves 0-3 play the GEMM role ("M"), waves 4-7 the attention role ("S"); a workgroup's waves are dealt to the four
// SIMDs in turn, so every SIMD holds one M wave and one S wave.  Each role is timed alone (the other half of the workgroup
// exits at once), then both together.
//
//   M variants   0: register-only v_mfma_f32_16x16x4_f32 on 8 accumulators (corun_probe's mfma_burn)
//                1: GEMM-shaped: per 16 MFMAs four ds_read_b128 (private LDS region), one 16-byte global load of an L2-resident
//                   weight row and its bf16 unpack (8 VALU) — the instruction mix of gemm_tile_kernel's block, no barrier
//   S variants   register-staged streaming as attn_kernel does it (two sets of UNR x 16 B per lane in flight; per 16 B the
//                bf16 unpack + 8 fmaf of the score phase), optional s_setprio
//   I            "intra-wave": every wave does both, interleaved by hand — per 16 MFMAs one 16-byte stream load is issued and
//                one earlier one consumed (DESIGN.md §9.4's proposal)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP %s @%d\n", hipGetErrorString(e_), __LINE__); exit(2);} } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ unsigned long long g_ts[4];
__device__ unsigned long long g_clk[2 * 512];  // per workgroup: shader cycles and 100 MHz ticks spent in the M role  // S first-in, S last-out, M first-in, M last-out (100 MHz)

struct Args {
  const char* src;     // streamed bytes
  const uint16_t* wl;  // small weight buffer (L2-resident)
  float* out;
  int items_per_wave;  // 1 KiB items per S wave
  int m_iters;         // 16-MFMA groups x 8 per M wave
  int run_s, run_m, prio_s, prio_m;
};

template <int UNR, bool LITE = false>
__device__ __forceinline__ float role_stream(const Args& a, int swave, int nswaves_per_wg) {
  const int lane = threadIdx.x & 63;
  const char* base = a.src + ((size_t)blockIdx.x * nswaves_per_wg + swave) * (size_t)a.items_per_wave * 1024 + lane * 16;
  const int last = a.items_per_wave - 1;
  uint4 ra[UNR], rb[UNR];
  float q[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) q[i] = 1.0f + 0.01f * (lane + i);
  float acc = 0.f;
#define ISSUE(R, I0) _Pragma("unroll") for (int u = 0; u < UNR; ++u) { \
    const dsm4 t = __builtin_nontemporal_load(reinterpret_cast<const dsm4*>(base + (size_t)min((I0) + u, last) * 1024)); R[u] = make_uint4(t[0], t[1], t[2], t[3]); }
#define USE(R, I0) _Pragma("unroll") for (int u = 0; u < UNR; ++u) { \
    if (LITE) { acc += __uint_as_float((R[u].x ^ R[u].y) ^ (R[u].z ^ R[u].w)); continue; } \
    float p = 0.f; \
    p = fmaf(q[0], __uint_as_float(R[u].x << 16), p); p = fmaf(q[1], __uint_as_float(R[u].x & 0xFFFF0000u), p); \
    p = fmaf(q[2], __uint_as_float(R[u].y << 16), p); p = fmaf(q[3], __uint_as_float(R[u].y & 0xFFFF0000u), p); \
    p = fmaf(q[4], __uint_as_float(R[u].z << 16), p); p = fmaf(q[5], __uint_as_float(R[u].z & 0xFFFF0000u), p); \
    p = fmaf(q[6], __uint_as_float(R[u].w << 16), p); p = fmaf(q[7], __uint_as_float(R[u].w & 0xFFFF0000u), p); \
    p += __shfl_xor(p, 8, 64); p += __shfl_xor(p, 4, 64); p += __shfl_xor(p, 2, 64); p += __shfl_xor(p, 1, 64); \
    if ((I0) + u <= last) acc += p; }
  typedef unsigned int dsm4 __attribute__((ext_vector_type(4)));
  int i = 0;
  ISSUE(ra, 0)
  while (i < a.items_per_wave) {
    ISSUE(rb, i + UNR)
    USE(ra, i)
    i += UNR;
    if (i >= a.items_per_wave) break;
    ISSUE(ra, i + UNR)
    USE(rb, i)
    i += UNR;
  }
#undef ISSUE
#undef USE
  return acc;
}

template <int MV, int GAP = 0>
__device__ __forceinline__ float role_mfma(const Args& a, int mwave, const float* lds) {
  const int lane = threadIdx.x & 63;
  f32x4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float s = 0.f;
  if (MV == 2) {  // bf16 matrix core, same number of instructions at 16 cycles each: twice the iterations
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    bf16x8 av, bv;
#pragma unroll
    for (int i = 0; i < 8; ++i) { av[i] = (__bf16)(1.0f + 0.01f * (lane + i)); bv[i] = (__bf16)(0.5f + 0.01f * (lane - i)); }
    for (int it = 0; it < 2 * a.m_iters; ++it) {
#pragma unroll
      for (int u = 0; u < 16; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc[i], 0, 0, 0);
    }
  } else if (MV == 3) {  // f32 VALU fma burn (v_pk_fma or v_fma): is the "f32 MFMA" the VALU in disguise?
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = 0.001f * (lane + i);
    const float m = 1.0001f, c = 0.0001f;
    for (int it = 0; it < a.m_iters * 64; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = fmaf(x[i], m, c);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i];
  } else if (MV == 0) {
    const float av = 1.2345f + lane * 1e-3f, bv = 0.9876f + lane * 1e-3f;
    for (int it = 0; it < a.m_iters; ++it) {
#pragma unroll
      for (int u = 0; u < 16; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[i], 0, 0, 0);
    }
  } else {
    // per step: 16 MFMAs (2 accumulators x 8 k-steps), 4 ds_read_b128, 1 global 16-byte load + unpack, as one block of
    // gemm_tile_kernel<MT = 2, NT = 1>; 8 steps per iteration so that the work per iteration equals MV == 0's
    const float* fp = lds + mwave * 2048 + (lane & 15) * 32 + 8 * (lane >> 4);
    const uint16_t* wp = a.wl + (size_t)((blockIdx.x * 4 + mwave) & 255) * 4096 + lane * 8;
    uint4 wnext = *reinterpret_cast<const uint4*>(wp);
    for (int it = 0; it < a.m_iters; ++it) {
#pragma unroll
      for (int st = 0; st < 8; ++st) {
        const uint4 w = wnext;
        wnext = *reinterpret_cast<const uint4*>(wp + ((it * 8 + st + 1) & 7) * 512);
        float wa[8];
        wa[0] = __uint_as_float(w.x << 16); wa[1] = __uint_as_float(w.x & 0xFFFF0000u);
        wa[2] = __uint_as_float(w.y << 16); wa[3] = __uint_as_float(w.y & 0xFFFF0000u);
        wa[4] = __uint_as_float(w.z << 16); wa[5] = __uint_as_float(w.z & 0xFFFF0000u);
        wa[6] = __uint_as_float(w.w << 16); wa[7] = __uint_as_float(w.w & 0xFFFF0000u);
        float xb[2][8];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const float4 f0 = *reinterpret_cast<const float4*>(fp + mt * 512 + (st & 1) * 1024 / 2);
          const float4 f1 = *reinterpret_cast<const float4*>(fp + mt * 512 + (st & 1) * 1024 / 2 + 4);
          xb[mt][0] = f0.x; xb[mt][1] = f0.y; xb[mt][2] = f0.z; xb[mt][3] = f0.w;
          xb[mt][4] = f1.x; xb[mt][5] = f1.y; xb[mt][6] = f1.z; xb[mt][7] = f1.w;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) acc[mt + 2 * (st & 3)] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[k], xb[mt][k], acc[mt + 2 * (st & 3)], 0, 0, 0);
        // yield: let the SIMD's other wave issue (GAP x 64 clocks per 16 MFMAs = 512 clocks)
        if (GAP == 1) __builtin_amdgcn_s_sleep(1);
        if (GAP == 2) __builtin_amdgcn_s_sleep(2);
        if (GAP == 4) __builtin_amdgcn_s_sleep(4);
        if (GAP == 11 && (st & 1)) __builtin_amdgcn_s_sleep(1);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  return s;
}

// roles by wave: waves 0-3 M, waves 4-7 S
template <int MV, int UNR, bool LITE = false, int GAP = 0>
__global__ __launch_bounds__(512, 1) void roles_kernel(Args a) {
  __shared__ __attribute__((aligned(16))) float lds[4 * 2048];
  for (int i = threadIdx.x; i < 4 * 2048; i += 512) lds[i] = 0.001f * (i & 1023);
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (wave < 4) {
    if (!a.run_m) return;
    if (a.prio_m) __builtin_amdgcn_s_setprio(1);
    if (lane == 0) atomicMin(&g_ts[2], (unsigned long long)wall_clock64());
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    const float s = role_mfma<MV, GAP>(a, wave, lds);
    if (s == 12345.678f) a.out[1] = s;
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0 && wave == 0 && blockIdx.x < 512) { g_clk[2 * blockIdx.x] = c1 - c0; g_clk[2 * blockIdx.x + 1] = r1 - r0; }
    if (lane == 0) atomicMax(&g_ts[3], (unsigned long long)wall_clock64());
  } else {
    if (!a.run_s) return;
    if (a.prio_s) __builtin_amdgcn_s_setprio(3);
    if (lane == 0) atomicMin(&g_ts[0], (unsigned long long)wall_clock64());
    const float s = role_stream<UNR, LITE>(a, wave - 4, 4);
    if (s == 12345.678f) a.out[0] = s;
    if (lane == 0) atomicMax(&g_ts[1], (unsigned long long)wall_clock64());
  }
}

// intra-wave: 4 waves per workgroup, each does the M work of variant 1 and streams with the loads and their use
// interleaved by hand between the MFMA groups: per step (16 MFMAs) SPS stream loads are issued and SPS consumed, DEPTH steps later
template <int SPS, int DEPTH>
__global__ __launch_bounds__(256, 2) void intra_kernel(Args a) {
  __shared__ __attribute__((aligned(16))) float lds[4 * 2048];
  for (int i = threadIdx.x; i < 4 * 2048; i += 256) lds[i] = 0.001f * (i & 1023);
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) atomicMin(&g_ts[0], (unsigned long long)wall_clock64());
  typedef unsigned int dsm4 __attribute__((ext_vector_type(4)));
  const char* base = a.src + ((size_t)blockIdx.x * 4 + wave) * (size_t)a.items_per_wave * 1024 + lane * 16;
  const int last = a.items_per_wave - 1;
  f32x4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float q[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) q[i] = 1.0f + 0.01f * (lane + i);
  float sacc = 0.f;
  const float* fp = lds + wave * 2048 + (lane & 15) * 32 + 8 * (lane >> 4);
  const uint16_t* wp = a.wl + (size_t)((blockIdx.x * 4 + wave) & 255) * 4096 + lane * 8;
  uint4 wnext = *reinterpret_cast<const uint4*>(wp);
  constexpr int RING = SPS * DEPTH;
  uint4 ring[RING];
#pragma unroll
  for (int u = 0; u < RING; ++u) {
    const dsm4 t = __builtin_nontemporal_load(reinterpret_cast<const dsm4*>(base + (size_t)min(u, last) * 1024));
    ring[u] = make_uint4(t[0], t[1], t[2], t[3]);
  }
  int item = 0;  // next item to consume
  // total steps: enough for both; the stream part stops issuing real items past `last` (clamped re-loads), the MFMA part runs m_iters * 8 steps
  const int steps_m = a.run_m ? a.m_iters * 8 : 0;
  const int steps_s = a.run_s ? (a.items_per_wave + SPS - 1) / SPS : 0;
  const int steps = max(steps_m, steps_s);
  for (int s0 = 0; s0 < steps; s0 += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int st = s0 + d;
      if (st < steps_s) {
#pragma unroll
        for (int u = 0; u < SPS; ++u) {
          const uint4 r = ring[d * SPS + u];
          float p = 0.f;
          p = fmaf(q[0], __uint_as_float(r.x << 16), p); p = fmaf(q[1], __uint_as_float(r.x & 0xFFFF0000u), p);
          p = fmaf(q[2], __uint_as_float(r.y << 16), p); p = fmaf(q[3], __uint_as_float(r.y & 0xFFFF0000u), p);
          p = fmaf(q[4], __uint_as_float(r.z << 16), p); p = fmaf(q[5], __uint_as_float(r.z & 0xFFFF0000u), p);
          p = fmaf(q[6], __uint_as_float(r.w << 16), p); p = fmaf(q[7], __uint_as_float(r.w & 0xFFFF0000u), p);
          p += __shfl_xor(p, 8, 64); p += __shfl_xor(p, 4, 64); p += __shfl_xor(p, 2, 64); p += __shfl_xor(p, 1, 64);
          if (item + u <= last) sacc += p;
          const dsm4 t = __builtin_nontemporal_load(reinterpret_cast<const dsm4*>(base + (size_t)min(item + u + RING, last) * 1024));
          ring[d * SPS + u] = make_uint4(t[0], t[1], t[2], t[3]);
        }
        item += SPS;
      }
      if (st < steps_m) {
        const uint4 w = wnext;
        wnext = *reinterpret_cast<const uint4*>(wp + ((st + 1) & 7) * 512);
        float wa[8];
        wa[0] = __uint_as_float(w.x << 16); wa[1] = __uint_as_float(w.x & 0xFFFF0000u);
        wa[2] = __uint_as_float(w.y << 16); wa[3] = __uint_as_float(w.y & 0xFFFF0000u);
        wa[4] = __uint_as_float(w.z << 16); wa[5] = __uint_as_float(w.z & 0xFFFF0000u);
        wa[6] = __uint_as_float(w.w << 16); wa[7] = __uint_as_float(w.w & 0xFFFF0000u);
        float xb[2][8];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const float4 f0 = *reinterpret_cast<const float4*>(fp + mt * 512 + (d & 1) * 512);
          const float4 f1 = *reinterpret_cast<const float4*>(fp + mt * 512 + (d & 1) * 512 + 4);
          xb[mt][0] = f0.x; xb[mt][1] = f0.y; xb[mt][2] = f0.z; xb[mt][3] = f0.w;
          xb[mt][4] = f1.x; xb[mt][5] = f1.y; xb[mt][6] = f1.z; xb[mt][7] = f1.w;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) acc[mt + 2 * (d & 3)] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[k], xb[mt][k], acc[mt + 2 * (d & 3)], 0, 0, 0);
      }
    }
  }
  float s = sacc;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678f) a.out[0] = s;
  if (lane == 0) atomicMax(&g_ts[1], (unsigned long long)wall_clock64());
}

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

static void reset_ts() {
  const unsigned long long init[4] = {~0ull, 0ull, ~0ull, 0ull};
  CK(hipMemcpyToSymbol(HIP_SYMBOL(g_ts), init, sizeof init));
}
static void print_ts() {
  unsigned long long ts[4];
  CK(hipMemcpyFromSymbol(ts, HIP_SYMBOL(g_ts), sizeof ts));
  const unsigned long long o = ts[0] < ts[2] ? ts[0] : ts[2];
  printf("   [S %.0f..%.0f us, M %.0f..%.0f us]", ts[0] == ~0ull ? -1.0 : (ts[0] - o) / 100.0, (ts[1] > o ? ts[1] - o : 0) / 100.0,
         ts[2] == ~0ull ? -1.0 : (ts[2] - o) / 100.0, (ts[3] > o ? ts[3] - o : 0) / 100.0);
}

static void print_clk(int wgs) {
  static unsigned long long h[2 * 512];
  CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_clk), sizeof h));
  std::vector<double> ghz;
  for (int i = 0; i < wgs && i < 512; ++i) if (h[2 * i + 1]) ghz.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 0.1);
  if (ghz.empty()) return;
  std::sort(ghz.begin(), ghz.end());
  printf(" clk %.2f GHz", ghz[ghz.size() / 2]);
}
template <int MV, int UNR, bool LITE = false, int GAP = 0>
void roles_case(const char* name, Args a, int wgs, double bytes, double flops) {
  auto run = [&](int s, int m) {
    Args b = a; b.run_s = s; b.run_m = m;
    reset_ts();
    return time_ms([&] { reset_ts(); hipLaunchKernelGGL((roles_kernel<MV, UNR, LITE, GAP>), dim3(wgs), dim3(512), 0, 0, b); }, 3);
  };
  const double ts = run(1, 0), tm = run(0, 1);
  printf("%-44s S alone %.3f ms (%4.0f GB/s)  M alone %.3f ms (%5.1f TF)", name, ts, bytes / ts / 1e6, tm, flops / tm / 1e9);
  print_clk(wgs);
  const double tb = run(1, 1);
  printf("  both %.3f ms  (serial %.3f, ideal %.3f)", tb, ts + tm, std::max(ts, tm));
  print_clk(wgs);
  print_ts();
  printf("\n");
  fflush(stdout);
}

template <int SPS, int DEPTH>
void intra_case(const char* name, Args a, int wgs, double bytes, double flops) {
  auto run = [&](int s, int m) {
    Args b = a; b.run_s = s; b.run_m = m;
    reset_ts();
    return time_ms([&] { hipLaunchKernelGGL((intra_kernel<SPS, DEPTH>), dim3(wgs), dim3(256), 0, 0, b); }, 3);
  };
  const double ts = run(1, 0), tm = run(0, 1), tb = run(1, 1);
  printf("%-44s S alone %.3f ms (%4.0f GB/s)  M alone %.3f ms (%5.1f TF)  both %.3f ms  (serial %.3f, ideal %.3f)\n", name, ts, bytes / ts / 1e6,
         tm, flops / tm / 1e9, tb, ts + tm, std::max(ts, tm));
  fflush(stdout);
}

int main(int argc, char** argv) {
  const double gb = argc > 1 ? atof(argv[1]) : 3.2;
  char* src;
  const size_t bytes = (size_t)(gb * 1e9) / (2048 * 1024) * (2048 * 1024);
  CK(hipMalloc(&src, bytes + (1 << 20)));
  CK(hipMemset(src, 0x3c, bytes + (1 << 20)));
  uint16_t* wl;
  CK(hipMalloc(&wl, 256 * 4096 * 2 + 65536));
  CK(hipMemset(wl, 0x3c, 256 * 4096 * 2 + 65536));
  float* out;
  CK(hipMalloc(&out, 64));
  Args a{};
  a.src = src; a.wl = wl; a.out = out;
  printf("fused_roles_probe: %.2f GB streamed per run\n", bytes / 1e9);
  for (int k = 1; k <= 2; ++k) {  // workgroups per CU
    const int wgs = 256 * k;
    a.items_per_wave = (int)(bytes / 1024 / (wgs * 4));
    const double flops = 0;  // filled per case
    (void)flops;
    // size M so that M alone ~ 0.55 ms: per wave per iteration 128 MFMAs x 32 cycles = 4096 cycles; k waves per SIMD share the pipe
    a.m_iters = 300 / k;
    const double fl = (double)wgs * 4 * a.m_iters * 128.0 * 2048.0;
    char nm[128];
    snprintf(nm, sizeof nm, "roles %d wg/CU, M0 burn, S UNR8", k);
    a.prio_s = 0; a.prio_m = 0;
    roles_case<0, 8>(nm, a, wgs, (double)bytes, fl);
    snprintf(nm, sizeof nm, "roles %d wg/CU, M0 burn, S UNR8 prio3", k);
    a.prio_s = 1;
    roles_case<0, 8>(nm, a, wgs, (double)bytes, fl);
    snprintf(nm, sizeof nm, "roles %d wg/CU, M1 gemm-mix, S UNR8", k);
    a.prio_s = 0;
    roles_case<1, 8>(nm, a, wgs, (double)bytes, fl);
    snprintf(nm, sizeof nm, "roles %d wg/CU, M1 gemm-mix, S UNR8 prio3", k);
    a.prio_s = 1;
    roles_case<1, 8>(nm, a, wgs, (double)bytes, fl);
    snprintf(nm, sizeof nm, "roles %d wg/CU, M1 prio1, S UNR8 prio0", k);
    a.prio_s = 0; a.prio_m = 1;
    roles_case<1, 8>(nm, a, wgs, (double)bytes, fl);
    a.prio_m = 0;
    snprintf(nm, sizeof nm, "roles %d wg/CU, M1 gemm-mix, S UNR4", k);
    roles_case<1, 4>(nm, a, wgs, (double)bytes, fl);
  }
  {
    const int wgs = 256;
    a.items_per_wave = (int)(bytes / 1024 / (wgs * 4));
    a.m_iters = 300;
    const double fl = (double)wgs * 4 * a.m_iters * 128.0 * 2048.0;
    a.prio_s = a.prio_m = 0;
    roles_case<1, 8, false, 11>("roles 1 wg/CU, M1 sleep1/32 MFMAs, S UNR8", a, wgs, (double)bytes, fl);
    roles_case<1, 8, false, 1>("roles 1 wg/CU, M1 sleep1/16 MFMAs, S UNR8", a, wgs, (double)bytes, fl);
    roles_case<1, 8, false, 2>("roles 1 wg/CU, M1 sleep2/16 MFMAs, S UNR8", a, wgs, (double)bytes, fl);
    roles_case<1, 8, false, 4>("roles 1 wg/CU, M1 sleep4/16 MFMAs, S UNR8", a, wgs, (double)bytes, fl);
    roles_case<1, 8, true, 1>("roles 1 wg/CU, M1 sleep1/16 MFMAs, S lite", a, wgs, (double)bytes, fl);
    a.prio_s = 1;
    roles_case<1, 8, false, 1>("roles 1 wg/CU, M1 sleep1/16, S UNR8 prio3", a, wgs, (double)bytes, fl);
    roles_case<1, 8, false, 2>("roles 1 wg/CU, M1 sleep2/16, S UNR8 prio3", a, wgs, (double)bytes, fl);
    a.prio_s = 0;
    roles_case<0, 8, true>("roles 1 wg/CU, M0 f32 burn, S lite (no VALU)", a, wgs, (double)bytes, fl);
    roles_case<1, 8, true>("roles 1 wg/CU, M1 gemm-mix, S lite (no VALU)", a, wgs, (double)bytes, fl);
    roles_case<2, 8, false>("roles 1 wg/CU, M2 bf16 MFMA burn, S UNR8", a, wgs, (double)bytes, fl * 16);
    roles_case<2, 8, true>("roles 1 wg/CU, M2 bf16 MFMA burn, S lite", a, wgs, (double)bytes, fl * 16);
    a.m_iters = 150;
    roles_case<3, 8, false>("roles 1 wg/CU, M3 f32 VALU fma burn, S UNR8", a, wgs, (double)bytes, (double)wgs * 4 * a.m_iters * 64 * 128.0 * 128.0);
    roles_case<3, 8, true>("roles 1 wg/CU, M3 f32 VALU fma burn, S lite", a, wgs, (double)bytes, (double)wgs * 4 * a.m_iters * 64 * 128.0 * 128.0);
  }
  // intra-wave: 512 workgroups of 256 threads (2 per CU = 2 waves per SIMD), each wave both roles
  {
    const int wgs = 512;
    a.items_per_wave = (int)(bytes / 1024 / (wgs * 4));
    a.m_iters = 150;
    const double fl = (double)wgs * 4 * a.m_iters * 128.0 * 2048.0;
    a.prio_s = a.prio_m = 0;
    intra_case<1, 8>("intra 2 wg/CU, 1 load/step, depth 8", a, wgs, (double)bytes, fl);
    intra_case<2, 4>("intra 2 wg/CU, 2 loads/step, depth 4", a, wgs, (double)bytes, fl);
    intra_case<2, 8>("intra 2 wg/CU, 2 loads/step, depth 8", a, wgs, (double)bytes, fl);
    intra_case<4, 4>("intra 2 wg/CU, 4 loads/step, depth 4", a, wgs, (double)bytes, fl);
  }
  return 0;
}
