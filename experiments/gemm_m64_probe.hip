// gemm_m64_probe.hip — where do the microseconds of a 64-row bx3 GEMM go? (r03)
//
// At B = 64 (one stream group, M = 64) the four LM GEMMs of a layer stream 102 MB of bf16 weights; at HBM speed that is
// 13 us, the engine's gemm_bx3_kernel launches take 56 us (rocprofv3, profiles/r03): QKV 13.6, out_proj 10.0, gate 19.8,
// ff_out 13.1 — and neither pre-split activation planes nor an eight-block weight window moved them.  This probe rebuilds
// that kernel in stages and in a few alternative tilings, to be run under `rocprofv3 --kernel-trace --stats`
// (every variant is its own template instance, i.e. its own line of the stats table; weights rotate through NBUF buffers
// larger than the Infinity Cache together):
//   STAGE 0  weights only: each wave loads its blocks (16 B per lane and block) and xors them into one store
//   STAGE 1  + activations: f32 loads, three-way split, LDS planes, barrier per block
//   STAGE 2  + the MFMAs (result stored only under an impossible condition)
//   STAGE 3  + the split-K slab store (the full kernel)
// tilings: NT n-tiles per wave (64 * NT columns per workgroup), KCH consecutive 256-wide chunks per workgroup (summed in
// order in registers, slabs shrink by KCH), WPB waves per workgroup.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP %s @%d\n", hipGetErrorString(e_), __LINE__); exit(2);} } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3(float x, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
  const uint32_t u = __float_as_uint(x);
  const float r1 = x - __uint_as_float(u & 0xFFFF0000u);
  const uint32_t u1 = __float_as_uint(r1);
  const float r2 = r1 - __uint_as_float(u1 & 0xFFFF0000u);
  hi = u >> 16; mid = u1 >> 16; lo = __float_as_uint(r2) >> 16;
}

// M = 64 (MT = 4).  grid (N / (64 NT), K / (256 KCH)); slab [chunk group][64][N]
template <int STAGE, int NT, int KCH>
__global__ __launch_bounds__(256, 2) void probe_kernel(const float* __restrict__ X, const uint16_t* __restrict__ W,
                                                       float* __restrict__ ws, int N, int K, int never) {
  constexpr int MT = 4;
  __shared__ __attribute__((aligned(16))) uint16_t Xp[2][3][16 * MT][32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
  const int n_base = blockIdx.x * 64 * NT + 16 * wave;
  const uint16_t* wrow[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) wrow[nt] = W + (size_t)(n_base + 64 * nt + r) * K + 8 * q;
  const int row0 = tid >> 3, part = tid & 7;
  const float* xsrc0 = X + (size_t)row0 * K + 4 * part;
  const float* xsrc1 = X + (size_t)(row0 + 32) * K + 4 * part;
  const int kb0 = blockIdx.y * 8 * KCH, kb1 = kb0 + 8 * KCH;
  f32x4 acc[NT][MT], tot[NT][MT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) { acc[nt][mt] = (f32x4){0.f, 0.f, 0.f, 0.f}; tot[nt][mt] = acc[nt][mt]; }
  auto stage = [&](int buf, float4 v, int row) {
    uint32_t h[4], m[4], l[4];
    split3(v.x, h[0], m[0], l[0]); split3(v.y, h[1], m[1], l[1]); split3(v.z, h[2], m[2], l[2]); split3(v.w, h[3], m[3], l[3]);
    const int unit = ((part >> 1) ^ ((row >> 1) & 3)), off = row * 32 + unit * 8 + (part & 1) * 4;
    *reinterpret_cast<uint2*>(&Xp[buf][0][0][0] + off) = make_uint2(l[0] | (l[1] << 16), l[2] | (l[3] << 16));
    *reinterpret_cast<uint2*>(&Xp[buf][1][0][0] + off) = make_uint2(m[0] | (m[1] << 16), m[2] | (m[3] << 16));
    *reinterpret_cast<uint2*>(&Xp[buf][2][0][0] + off) = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
  };
  constexpr int DW = 8;
  uint4 wq[DW][NT];
#pragma unroll
  for (int u = 0; u < DW; ++u)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) wq[u][nt] = *reinterpret_cast<const uint4*>(wrow[nt] + 32 * (kb0 + u));
  float4 xa = make_float4(0, 0, 0, 0), xb = xa;
  if (STAGE >= 1) { xa = *reinterpret_cast<const float4*>(xsrc0 + 32 * kb0); xb = *reinterpret_cast<const float4*>(xsrc1 + 32 * kb0); }
  uint4 sink = make_uint4(0, 0, 0, 0);
#pragma clang loop unroll(disable)
  for (int g0 = kb0; g0 < kb1; g0 += DW) {
#pragma unroll
    for (int u = 0; u < DW; ++u) {
      const int g = g0 + u, buf = u & 1;
      if (STAGE >= 1) { stage(buf, xa, row0); stage(buf, xb, row0 + 32); }
      bf16x8 wa[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const uint4 w = wq[u][nt];
        if (STAGE < 2) { sink.x ^= w.x; sink.y ^= w.y; sink.z ^= w.z; sink.w ^= w.w; }
        wa[nt][0] = (short)(w.x & 0xFFFF); wa[nt][1] = (short)(w.x >> 16); wa[nt][2] = (short)(w.y & 0xFFFF); wa[nt][3] = (short)(w.y >> 16);
        wa[nt][4] = (short)(w.z & 0xFFFF); wa[nt][5] = (short)(w.z >> 16); wa[nt][6] = (short)(w.w & 0xFFFF); wa[nt][7] = (short)(w.w >> 16);
      }
      const int gn = g + 1 < kb1 ? g + 1 : g;
      if (STAGE >= 1) { xa = *reinterpret_cast<const float4*>(xsrc0 + 32 * gn); xb = *reinterpret_cast<const float4*>(xsrc1 + 32 * gn); }
      if (KCH > 1) {
        const int gw = g + DW < kb1 ? g + DW : kb1 - 1;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) wq[u][nt] = *reinterpret_cast<const uint4*>(wrow[nt] + 32 * gw);
      }
      if (STAGE >= 1) __syncthreads();
      if (STAGE >= 2) {
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            const int row = 16 * mt + r, unit = q ^ ((row >> 1) & 3);
            const bf16x8 xf = *reinterpret_cast<const bf16x8*>(&Xp[buf][p][0][0] + row * 32 + unit * 8);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[nt], xf, acc[nt][mt], 0, 0, 0);
          }
      } else if (STAGE == 1) {
        const uint4 t = *reinterpret_cast<const uint4*>(&Xp[buf][lane % 3][0][0] + (16 * (lane & 3) + r) * 32 + q * 8);
        sink.x ^= t.x; sink.y ^= t.y;
      }
    }
    // a 256-wide chunk is complete
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) { tot[nt][mt] = tot[nt][mt] + acc[nt][mt]; acc[nt][mt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  }
  if (STAGE < 2) {
    if ((sink.x ^ sink.y ^ sink.z ^ sink.w) == 0x12345u + (uint32_t)never) ws[tid] = 1.f;
    return;
  }
  if (STAGE == 2 && never == 0) return;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
      *reinterpret_cast<f32x4*>(ws + ((size_t)blockIdx.y * 64 + 16 * mt + r) * N + n_base + 64 * nt + 4 * q) = tot[nt][mt];
}

// v2: the activation window is as deep as the weight window and issued FIRST.  vmcnt retires in issue order: in the kernel
// above the activation block for g + 1 is requested after the weights for g + 7, so waiting for it drains the whole weight
// window — the effective look-ahead of every stream is one block (the ISA shows s_waitcnt vmcnt(1..2) in every block).
// D blocks of both streams in flight; split-K (KCH = 1): everything is requested up front, activations first.
template <int NT, int KCH, int D>
__global__ __launch_bounds__(256, 2) void probe2_kernel(const float* __restrict__ X, const uint16_t* __restrict__ W,
                                                        float* __restrict__ ws, int N, int K, int never) {
  constexpr int MT = 4;
  __shared__ __attribute__((aligned(16))) uint16_t Xp[2][3][16 * MT][32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
  const int n_base = blockIdx.x * 64 * NT + 16 * wave;
  const uint16_t* wrow[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) wrow[nt] = W + (size_t)(n_base + 64 * nt + r) * K + 8 * q;
  const int row0 = tid >> 3, part = tid & 7;
  const float* xsrc0 = X + (size_t)row0 * K + 4 * part;
  const float* xsrc1 = X + (size_t)(row0 + 32) * K + 4 * part;
  const int kb0 = blockIdx.y * 8 * KCH, kb1 = kb0 + 8 * KCH;
  f32x4 acc[NT][MT], tot[NT][MT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) { acc[nt][mt] = (f32x4){0.f, 0.f, 0.f, 0.f}; tot[nt][mt] = acc[nt][mt]; }
  auto stage = [&](int buf, float4 v, int row) {
    uint32_t h[4], m[4], l[4];
    split3(v.x, h[0], m[0], l[0]); split3(v.y, h[1], m[1], l[1]); split3(v.z, h[2], m[2], l[2]); split3(v.w, h[3], m[3], l[3]);
    const int unit = ((part >> 1) ^ ((row >> 1) & 3)), off = row * 32 + unit * 8 + (part & 1) * 4;
    *reinterpret_cast<uint2*>(&Xp[buf][0][0][0] + off) = make_uint2(l[0] | (l[1] << 16), l[2] | (l[3] << 16));
    *reinterpret_cast<uint2*>(&Xp[buf][1][0][0] + off) = make_uint2(m[0] | (m[1] << 16), m[2] | (m[3] << 16));
    *reinterpret_cast<uint2*>(&Xp[buf][2][0][0] + off) = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
  };
  float4 xq[D][2];
  uint4 wq[D][NT];
#pragma unroll
  for (int u = 0; u < D; ++u) {
    xq[u][0] = *reinterpret_cast<const float4*>(xsrc0 + 32 * (kb0 + u));
    xq[u][1] = *reinterpret_cast<const float4*>(xsrc1 + 32 * (kb0 + u));
  }
  __builtin_amdgcn_sched_barrier(0);  // the scheduler would sink most of these below the weight loads (register pressure)
#pragma unroll
  for (int u = 0; u < D; ++u)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) wq[u][nt] = *reinterpret_cast<const uint4*>(wrow[nt] + 32 * (kb0 + u));
  __builtin_amdgcn_sched_barrier(0);
#pragma clang loop unroll(disable)
  for (int g0 = kb0; g0 < kb1; g0 += D) {
#pragma unroll
    for (int u = 0; u < D; ++u) {
      const int g = g0 + u, buf = u & 1;
      stage(buf, xq[u][0], row0);
      stage(buf, xq[u][1], row0 + 32);
      bf16x8 wa[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const uint4 w = wq[u][nt];
        wa[nt][0] = (short)(w.x & 0xFFFF); wa[nt][1] = (short)(w.x >> 16); wa[nt][2] = (short)(w.y & 0xFFFF); wa[nt][3] = (short)(w.y >> 16);
        wa[nt][4] = (short)(w.z & 0xFFFF); wa[nt][5] = (short)(w.z >> 16); wa[nt][6] = (short)(w.w & 0xFFFF); wa[nt][7] = (short)(w.w >> 16);
      }
      if (8 * KCH > D) {
        const int gw = g + D < kb1 ? g + D : kb1 - 1;
        __builtin_amdgcn_sched_barrier(0);
        xq[u][0] = *reinterpret_cast<const float4*>(xsrc0 + 32 * gw);
        xq[u][1] = *reinterpret_cast<const float4*>(xsrc1 + 32 * gw);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) wq[u][nt] = *reinterpret_cast<const uint4*>(wrow[nt] + 32 * gw);
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const int row = 16 * mt + r, unit = q ^ ((row >> 1) & 3);
          const bf16x8 xf = *reinterpret_cast<const bf16x8*>(&Xp[buf][p][0][0] + row * 32 + unit * 8);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[nt], xf, acc[nt][mt], 0, 0, 0);
        }
      if ((g & 7) == 7) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) { tot[nt][mt] = tot[nt][mt] + acc[nt][mt]; acc[nt][mt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
      }
    }
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
      *reinterpret_cast<f32x4*>(ws + ((size_t)blockIdx.y * 64 + 16 * mt + r) * N + n_base + 64 * nt + 4 * q) = tot[nt][mt];
}

template <int NT, int KCH, int D>
void run2(int N, int K, const float* X, uint16_t* const* W, int nbuf, float* ws, hipStream_t st, int reps) {
  dim3 grid(N / (64 * NT), K / (256 * KCH));
  for (int i = 0; i < reps; ++i)
    hipLaunchKernelGGL((probe2_kernel<NT, KCH, D>), grid, dim3(256), 0, st, X, W[i % nbuf], ws, N, K, 0);
  CK(hipGetLastError());
}

struct Shape { const char* name; int N, K; };

template <int STAGE, int NT, int KCH>
void run(const Shape& sh, const float* X, uint16_t* const* W, int nbuf, float* ws, hipStream_t st, int reps) {
  dim3 grid(sh.N / (64 * NT), sh.K / (256 * KCH));
  for (int i = 0; i < reps; ++i)
    hipLaunchKernelGGL((probe_kernel<STAGE, NT, KCH>), grid, dim3(256), 0, st, X, W[i % nbuf], ws, sh.N, sh.K, 0);
  CK(hipGetLastError());
}

int main() {
  const Shape shapes[4] = {{"qkv", 6144, 2048}, {"out_proj", 2048, 2048}, {"gate", 11264, 2048}, {"ff_out", 2048, 5632}};
  const int M = 64, NBUF = 12, reps = 60;
  hipStream_t st;
  CK(hipStreamCreate(&st));
  for (const Shape& sh : shapes) {
    float* X; float* ws; uint16_t* W[NBUF];
    std::vector<float> hx((size_t)M * sh.K);
    for (size_t i = 0; i < hx.size(); ++i) hx[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
    CK(hipMalloc(&X, hx.size() * 4));
    CK(hipMemcpy(X, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&ws, (size_t)(sh.K / 256) * M * sh.N * 4));
    std::vector<uint16_t> hw((size_t)sh.N * sh.K);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = (uint16_t)(0x3C00u + ((i * 40503u) & 0x3FF));
    for (int b = 0; b < NBUF; ++b) { CK(hipMalloc(&W[b], hw.size() * 2)); CK(hipMemcpy(W[b], hw.data(), hw.size() * 2, hipMemcpyHostToDevice)); }
    CK(hipDeviceSynchronize());
    // name the shape in the trace: one marker kernel launch per shape is overkill — the grid sizes tell the shapes apart
    run<0, 1, 1>(sh, X, W, NBUF, ws, st, reps);
    run<1, 1, 1>(sh, X, W, NBUF, ws, st, reps);
    run<2, 1, 1>(sh, X, W, NBUF, ws, st, reps);
    run<3, 1, 1>(sh, X, W, NBUF, ws, st, reps);
    run<0, 2, 1>(sh, X, W, NBUF, ws, st, reps);
    run<3, 2, 1>(sh, X, W, NBUF, ws, st, reps);
    if (sh.K % 512 == 0) {
      run<0, 1, 2>(sh, X, W, NBUF, ws, st, reps);
      run<3, 1, 2>(sh, X, W, NBUF, ws, st, reps);
      run<3, 2, 2>(sh, X, W, NBUF, ws, st, reps);
    }
    if (sh.K % 1024 == 0) {
      run<3, 1, 4>(sh, X, W, NBUF, ws, st, reps);
    }
    run2<1, 1, 8>(sh.N, sh.K, X, W, NBUF, ws, st, reps);
    run2<2, 1, 8>(sh.N, sh.K, X, W, NBUF, ws, st, reps);
    if (sh.K % 512 == 0) {
      run2<1, 2, 8>(sh.N, sh.K, X, W, NBUF, ws, st, reps);
      run2<1, 2, 4>(sh.N, sh.K, X, W, NBUF, ws, st, reps);
      run2<2, 2, 4>(sh.N, sh.K, X, W, NBUF, ws, st, reps);
    }
    if (sh.K % 1024 == 0) {
      run2<1, 4, 8>(sh.N, sh.K, X, W, NBUF, ws, st, reps);
      run2<1, 4, 4>(sh.N, sh.K, X, W, NBUF, ws, st, reps);
    }
    if (sh.K % 2048 == 0) run2<1, 8, 4>(sh.N, sh.K, X, W, NBUF, ws, st, reps);
    CK(hipDeviceSynchronize());
    printf("%s done: N=%d K=%d weights %.1f MB\n", sh.name, sh.N, sh.K, sh.N * (double)sh.K * 2 / 1e6);
    for (int b = 0; b < NBUF; ++b) CK(hipFree(W[b]));
    CK(hipFree(X)); CK(hipFree(ws));
  }
  return 0;
}
