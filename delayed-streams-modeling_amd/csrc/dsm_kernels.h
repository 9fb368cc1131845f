// dsm_kernels.h — hand-written HIP kernels for gfx950 (MI355X / CDNA4).  Included by dsm_engine.hip.
//
// Every kernel follows the canonical reduction orders of dsm_numerics.h so that its f32
// results are bit-identical to the CPU oracle (oracle/dsm_oracle.c).  Wavefront = 64 lanes.
//
//   gemm_mfma_kernel   Y = X . W^T on v_mfma_f32_16x16x4_f32 (exact f32, a k-ordered fmaf chain:
//                      experiments/numerics_probe.hip).  One workgroup = S waves, wave w owns the
//                      K-chunk [512w, 512w+512); chunk partials are summed left to right through
//                      LDS.  Fused epilogues: bias / GELU / layer-scale / residual / ELU copy,
//                      QKV split + RoPE + ring-cache scatter, SiLU gate, RVQ distance + argmin.
//   row_norm_kernel    RMSNorm / LayerNorm, one wave per row, xor-butterfly reductions.
//   attn_kernel        softmax(q.K^T/sqrt(hd) + ring mask).V, one 4-wave workgroup per (slot, head),
//                      K/V streamed once from HBM with 16-byte loads, scores staged in LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "dsm_numerics.h"
#include "dsm_sampling.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short dsm_bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float wave_sum64(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = v + __shfl_xor(v, off, 64);
  return v;
}

// ------------------------------------------------------------------------------------------
// GEMM
// ------------------------------------------------------------------------------------------
enum { EPI_STORE = 0, EPI_QKV = 1, EPI_GATE = 2, EPI_RVQ = 3 };

struct RowMap {  // row m -> element offset: (m / rpb) * bstride + ((m % rpb) + toff) * ld
  long bstride;
  int rpb;
  int ld;
  int toff;
  __host__ __device__ long off(int m) const { return (long)(m / rpb) * bstride + (long)((m % rpb) + toff) * ld; }
};

struct GemmArgs {
  const float* X;
  RowMap xmap;
  const void* W;  // packed [Npad][Kpad], K zero-padded to a multiple of 32, rows to a multiple of 16
  int Kpad, K, N, M;
  int nt_stride;  // row distance between the NT n-tiles of one wave (16, or `hidden` for the gate)
  int wpacked;    // bf16 W is fragment-major [N / 16][Kpad / 32][64][8] (pack_linear, r04) instead of row-major [N][Kpad]
  int w_ntiles;   // 16-row tiles of W in all (chunk-major experiment, dsm_gemm_wk.h)
  int wg_cols;    // weight rows per workgroup (0 = 64).  128 with nt_stride 64: gemm_bx3_kernel's two-n-tile form for plain
                  // epilogues — wave w owns columns 16w.. and 64 + 16w.. of the workgroup's 128
  // EPI_STORE
  const float* bias;   // [N] or null
  const float* scale;  // [N] or null (LayerScale)
  const float* res;    // residual source or null
  RowMap rmap;
  float* Y;
  RowMap ymap;
  float* Y2;  // optional ELU(y) copy
  RowMap y2map;
  int act;  // 0 none, 1 gelu_erf (applied right after bias)
  int vec;  // 1: every row offset of Y / Y2 / res is a multiple of 4 floats and N % 4 == 0 -> 16-byte accesses
  // EPI_QKV
  int d, hd, H, T, ctx;
  void* kcache;
  void* vcache;               // [B][H][ctx][hd]
  const uint32_t* widx;       // [B*T] ring slot written by row (b,t)
  const float* rope_cs;       // [B*T][hd/2][2] (cos, sin) or null
  const uint8_t* active;      // [B]
  // EPI_RVQ
  float* pval;     // [n_tiles][M]
  uint32_t* pidx;  // [n_tiles][M]
  // split-K workspace of the tiled kernel: [chunks][M padded to 16][ws_ntiles*16] f32, row-major
  float* ws;
  int ws_ntiles;
  int defer_reduce;  // EPI_QKV, T = 1: leave the slabs to the attention kernel's prologue (AttnFused)
  int chunk_loop;    // > 1: one workgroup walks all K-chunks itself (grid.y = 1) and adds the chunk sums left to right in
                     // registers — no slabs, no reduce launch; chosen when the (n, m) tiles alone fill the chip
  // optional row norm fused behind an EPI_STORE GEMM whose N is d_model (gemm_reduce_rows_kernel)
  const float* norm_w;
  const float* norm_b;
  float* norm_out;
  float norm_eps;
  int norm_rms;
  int wk_hint;  // host side: prefer the whole-K-in-the-workgroup form (gemm_wk_kernel) for this EPI_STORE launch
  // optional row norm IN FRONT of the GEMM (gemm_wkn_kernel): X is the un-normalised stream, d = K
  const float* pre_norm_w;
  const float* pre_norm_b;
  float pre_norm_eps;
  int pre_norm_rms;
  // optional device-clock bracket of the launch (timeline diagnostics, dsm_prof_timeline): {first workgroup in, last out}
  unsigned long long* ts;
};

// Where lane (r, q) finds its 8 consecutive k of weight row n0 + r in the first 32-wide block, and how far apart blocks are
// (elements): row-major, or — bf16 weights since r04 — fragment-major (pack_linear).  n0 is a multiple of 16.
template <typename WT>
__device__ __forceinline__ const WT* dsm_wbase(const GemmArgs& a, const WT* W, int n0, int r, int q) {
  if (sizeof(WT) == 2 && a.wpacked) return W + ((long)(n0 >> 4) * (a.Kpad >> 5)) * 512 + (q * 16 + r) * 8;
  return W + (long)(n0 + r) * a.Kpad + 8 * q;
}
template <typename WT>
__device__ __forceinline__ int dsm_wstep(const GemmArgs& a) { return (sizeof(WT) == 2 && a.wpacked) ? 16 : 1; }  // x k (multiples of 32)

// first / last workgroups of a launch stamp the wall clock (dispatch is in index order; a few hundred atomics at most)
__device__ __forceinline__ void launch_stamp_begin(unsigned long long* ts) {
  if (!ts || threadIdx.x != 0) return;
  const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  if (lin < 128) atomicMin(&ts[0], wall_clock64());
}
__device__ __forceinline__ void launch_stamp_end(unsigned long long* ts) {
  if (!ts || threadIdx.x != 0) return;
  const unsigned total = gridDim.x * gridDim.y * gridDim.z;
  const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  if (lin + 512 >= total) atomicMax(&ts[1], wall_clock64());
}

template <typename WT>
__device__ __forceinline__ void load_w8(const WT* p, float (&o)[8]);
template <>
__device__ __forceinline__ void load_w8<uint16_t>(const uint16_t* p, float (&o)[8]) {
  uint4 v = *reinterpret_cast<const uint4*>(p);
  o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xFFFF0000u);
  o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xFFFF0000u);
  o[4] = __uint_as_float(v.z << 16); o[5] = __uint_as_float(v.z & 0xFFFF0000u);
  o[6] = __uint_as_float(v.w << 16); o[7] = __uint_as_float(v.w & 0xFFFF0000u);
}
template <>
__device__ __forceinline__ void load_w8<float>(const float* p, float (&o)[8]) {
  float4 a = *reinterpret_cast<const float4*>(p);
  float4 b = *reinterpret_cast<const float4*>(p + 4);
  o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}

// 8 consecutive cache elements kept as raw words until they are used (prefetch registers of the attention kernel)
template <typename KVT>
struct Raw8;
typedef unsigned int dsm_u32x4 __attribute__((ext_vector_type(4)));
typedef float dsm_f32x4v __attribute__((ext_vector_type(4)));
template <>
struct Raw8<uint16_t> {
  uint4 v;
  __device__ __forceinline__ void load(const uint16_t* p) { v = *reinterpret_cast<const uint4*>(p); }
  // non-temporal: a ring cache is read once per step and is far larger than the Infinity Cache — do not let it evict the
  // weights the other stream group is about to read again
  __device__ __forceinline__ void load_nt(const uint16_t* p) {
    const dsm_u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const dsm_u32x4*>(p));
    v = make_uint4(t[0], t[1], t[2], t[3]);
  }
  __device__ __forceinline__ void unpack(float (&o)[8]) const {
    o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xFFFF0000u);
    o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xFFFF0000u);
    o[4] = __uint_as_float(v.z << 16); o[5] = __uint_as_float(v.z & 0xFFFF0000u);
    o[6] = __uint_as_float(v.w << 16); o[7] = __uint_as_float(v.w & 0xFFFF0000u);
  }
};
template <>
struct Raw8<float> {
  float4 a, b;
  __device__ __forceinline__ void load(const float* p) {
    a = *reinterpret_cast<const float4*>(p);
    b = *reinterpret_cast<const float4*>(p + 4);
  }
  __device__ __forceinline__ void load_nt(const float* p) {
    const dsm_f32x4v t0 = __builtin_nontemporal_load(reinterpret_cast<const dsm_f32x4v*>(p));
    const dsm_f32x4v t1 = __builtin_nontemporal_load(reinterpret_cast<const dsm_f32x4v*>(p + 4));
    a = make_float4(t0[0], t0[1], t0[2], t0[3]);
    b = make_float4(t1[0], t1[1], t1[2], t1[3]);
  }
  __device__ __forceinline__ void unpack(float (&o)[8]) const {
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
  }
};

__device__ __forceinline__ void store_kv(uint16_t* p, float v) { *p = dsm_f32_to_bf16(v); }
__device__ __forceinline__ void store_kv(float* p, float v) { *p = v; }

__device__ __forceinline__ void store_kv4(uint16_t* p, const float (&o)[4]) {
  uint2 v;
  v.x = (uint32_t)dsm_f32_to_bf16(o[0]) | ((uint32_t)dsm_f32_to_bf16(o[1]) << 16);
  v.y = (uint32_t)dsm_f32_to_bf16(o[2]) | ((uint32_t)dsm_f32_to_bf16(o[3]) << 16);
  *reinterpret_cast<uint2*>(p) = v;
}
__device__ __forceinline__ void store_kv4(float* p, const float (&o)[4]) {
  *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
}

// ---- epilogues (shared by the three GEMM kernels).  A lane holds rows n..n+3 of column m of a 16x16 tile. ----
template <typename KVT, int EPI>
__device__ __forceinline__ void epi_store_qkv(const GemmArgs& a, f32x4 v, int m, int n) {
  if (m >= a.M || n >= a.N) return;
  const bool full = a.vec && (n + 3 < a.N);
  if (EPI == EPI_STORE) {
    float o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float t = v[i];
      if (n + i < a.N) {
        if (a.bias) t = t + a.bias[n + i];
        if (a.act == 1) t = dsm_gelu_erf(t);
        if (a.scale) t = t * a.scale[n + i];
      }
      o[i] = t;
    }
    if (a.res) {
      const float* rp = a.res + a.rmap.off(m) + n;
      if (full) {
        float4 rv = *reinterpret_cast<const float4*>(rp);
        o[0] = rv.x + o[0]; o[1] = rv.y + o[1]; o[2] = rv.z + o[2]; o[3] = rv.w + o[3];
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (n + i < a.N) o[i] = rp[i] + o[i];
      }
    }
    if (a.Y) {
      float* y = a.Y + a.ymap.off(m) + n;
      if (full) {
        *reinterpret_cast<float4*>(y) = make_float4(o[0], o[1], o[2], o[3]);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (n + i < a.N) y[i] = o[i];
      }
    }
    if (a.Y2) {
      float* y2 = a.Y2 + a.y2map.off(m) + n;
      if (full) {
        *reinterpret_cast<float4*>(y2) = make_float4(dsm_elu(o[0]), dsm_elu(o[1]), dsm_elu(o[2]), dsm_elu(o[3]));
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (n + i < a.N) y2[i] = dsm_elu(o[i]);
      }
    }
  } else {  // EPI_QKV
    // n in [0, 3d): part 0 = q, 1 = k, 2 = v; (b,t,3,H,hd) layout — core/batched_transformer.rs:77-82
    const int part = n / a.d, c = n - part * a.d, h = c / a.hd, i0 = c - h * a.hd;
    const int b = m / a.T;
    float o[4] = {v[0], v[1], v[2], v[3]};
    if (part < 2 && a.rope_cs) {  // rope_i on interleaved pairs — core/transformer.rs:373-377
      const float4 cs = *reinterpret_cast<const float4*>(a.rope_cs + ((long)m * (a.hd / 2) + (i0 >> 1)) * 2);
      const float co[2] = {cs.x, cs.z}, si[2] = {cs.y, cs.w};
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        float x0 = v[2 * p], x1 = v[2 * p + 1];
        float t0 = x0 * co[p], t1 = x1 * si[p], t2 = x0 * si[p], t3 = x1 * co[p];
        o[2 * p] = t0 - t1;
        o[2 * p + 1] = t2 + t3;
      }
    }
    if (part == 0) {
      *reinterpret_cast<float4*>(a.Y + (long)m * a.d + c) = make_float4(o[0], o[1], o[2], o[3]);
    } else if (a.active[b]) {  // inactive slots: the reference scatters garbage that is never read
      KVT* cache = reinterpret_cast<KVT*>(part == 1 ? a.kcache : a.vcache);
      store_kv4(cache + (((long)b * a.H + h) * a.ctx + a.widx[m]) * a.hd + i0, o);
    }
  }
}

// Mlp::Gating — core/batched_transformer.rs:170-176: silu(first half) * second half
__device__ __forceinline__ void epi_gate(const GemmArgs& a, f32x4 g, f32x4 u, int m, int n) {
  if (m >= a.M || n >= a.N) return;
  float o[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = dsm_silu(g[i]) * u[i];
  float* y = a.Y + a.ymap.off(m) + n;
  if (a.vec && n + 3 < a.N) {
    *reinterpret_cast<float4*>(y) = make_float4(o[0], o[1], o[2], o[3]);
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (n + i < a.N) y[i] = o[i];
  }
}

// EuclideanCodebook::encode_slow — core/quantization.rs:122-131: dist = c2 - dot, argmin over the 16 codebook
// rows of this tile (first occurrence on ties); all 64 lanes of the wave must call it.
__device__ __forceinline__ void epi_rvq(const GemmArgs& a, f32x4 v, int m, int n, int tile16, int q) {
  float bv = DSM_INF_F;
  uint32_t bi = 0xFFFFFFFFu;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float dist = a.bias[n + i] - v[i];
    if (n + i < a.N && (dist < bv || (dist == bv && (uint32_t)(n + i) < bi))) { bv = dist; bi = (uint32_t)(n + i); }
  }
#pragma unroll
  for (int off = 16; off <= 32; off <<= 1) {
    float ov = __shfl_xor(bv, off, 64);
    uint32_t oi = __shfl_xor(bi, off, 64);
    if (ov < bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
  }
  if (q == 0 && m < a.M) {
    a.pval[(long)tile16 * a.M + m] = bv;
    a.pidx[(long)tile16 * a.M + m] = bi;
  }
}

// ---- generic GEMM (any K, any X alignment): one workgroup = S waves, wave w accumulates K-chunks w, w+S, ...
// and parks each chunk's partial tile in LDS; the partials are summed left to right.  Used for the few layers
// the tiled kernel below cannot take (K % 32 != 0 or unaligned rows: the 1-channel input conv, tiny test models).
template <typename WT, typename KVT, int MT, int NT, int EPI, bool XALIGNED, bool BX3 = false>
__global__ void gemm_mfma_kernel(GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds_part[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, S = blockDim.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int m_base = blockIdx.y * (16 * MT);
  const int n_base = blockIdx.x * ((EPI == EPI_GATE) ? 16 : 16 * NT);
  const WT* W = reinterpret_cast<const WT*>(a.W);
  constexpr int TILES = NT * MT;

  const WT* wrow[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) wrow[nt] = dsm_wbase<WT>(a, W, n_base + nt * a.nt_stride, r, q);
  const int wst = dsm_wstep<WT>(a);
  const float* xrow[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    int m = m_base + 16 * mt + r;
    m = m < a.M ? m : a.M - 1;  // padded rows re-read the last real row; their results are discarded
    xrow[mt] = a.X + a.xmap.off(m) + 8 * q;
  }
  f32x4 acc[NT][MT];
  const int chunks = (a.Kpad + DSM_KC - 1) / DSM_KC;

  for (int c = wave; c < chunks; c += S) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int k0 = c * DSM_KC;
    const int k1 = min(k0 + DSM_KC, a.Kpad);
    for (int kb = k0; kb < k1; kb += 32) {
      float wa[NT][8], xb[MT][8];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) load_w8<WT>(wrow[nt] + (long)kb * wst, wa[nt]);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        if (XALIGNED) {
          load_w8<float>(xrow[mt] + kb, xb[mt]);
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) xb[mt][j] = (kb + 8 * q + j < a.K) ? xrow[mt][kb + j] : 0.0f;
        }
      }
      if (BX3) {  // dot_mode 1: three bf16 pieces of the activations against the bf16 weights (gemm_bx3_kernel's order)
        dsm_bf16x8 wv[NT], xp[3][MT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int j = 0; j < 8; ++j) wv[nt][j] = (short)(__float_as_uint(wa[nt][j]) >> 16);  // the weights are bf16 values
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            uint16_t h, m, l;
            dsm_split3(xb[mt][j], &h, &m, &l);
            xp[0][mt][j] = (short)l; xp[1][mt][j] = (short)m; xp[2][mt][j] = (short)h;
          }
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv[nt], xp[p][mt], acc[nt][mt], 0, 0, 0);
        continue;
      }
#pragma unroll
      for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
            acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[nt][s], xb[mt][s], acc[nt][mt], 0, 0, 0);
    }
    if (chunks > 1) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          *reinterpret_cast<f32x4*>(&lds_part[(((c * TILES) + nt * MT + mt) * 64 + lane) * 4]) = acc[nt][mt];
    }
  }
  // split-K combine ((c0 + c1) + c2) + ...  GATE needs both tiles of an m-tile in one wave: wave w' owns
  // m-tiles w', w'+S, ...; otherwise tile (nt, mt) is owned by wave (nt*MT + mt) % S.
  const bool multi = chunks > 1;
  if (multi) {
    __syncthreads();
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int owner = (EPI == EPI_GATE) ? (mt % S) : ((nt * MT + mt) % S);
        if (owner != wave) continue;
        f32x4 tot = *reinterpret_cast<f32x4*>(&lds_part[((nt * MT + mt) * 64 + lane) * 4]);
        for (int c = 1; c < chunks; ++c) {
          f32x4 p = *reinterpret_cast<f32x4*>(&lds_part[(((c * TILES) + nt * MT + mt) * 64 + lane) * 4]);
          tot = tot + p;
        }
        acc[nt][mt] = tot;
      }
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = m_base + 16 * mt + r;
    if (EPI == EPI_GATE) {
      if (multi && (mt % S) != wave) continue;
      epi_gate(a, acc[0][mt], acc[NT - 1][mt], m, n_base + 4 * q);
      continue;
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      if (multi && ((nt * MT + mt) % S) != wave) continue;  // wave-uniform
      const int n = n_base + nt * a.nt_stride + 4 * q;
      if (EPI == EPI_RVQ)
        epi_rvq(a, acc[nt][mt], m, n, (n_base + nt * a.nt_stride) >> 4, q);
      else
        epi_store_qkv<KVT, EPI>(a, acc[nt][mt], m, n);
    }
  }
}

// ---- tiled GEMM (fast path: K % 32 == 0, 16-byte aligned X rows).
// Workgroup = 4 waves; tile = 64 weight rows (wave w owns 16-row n-tile w; for the gate also the matching
// "up" tile at +nt_stride) x 16*MT activation rows x ONE K-chunk (blockIdx.y).  The 32-wide activation block
// [16*MT][32] f32 is fetched once per workgroup with coalesced full-line loads, staged in LDS and shared by the four
// waves; weights stream straight to registers.  LDS image (r02): unpadded 32-float rows whose eight 16-byte units are
// XOR-swizzled by bits 1 and 3 of the row (dsm_xs_sw) — the lane groups that one ds_read_b128 services together are
// {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... (MI355X_MICROARCH.md, LDS), i.e. fragment rows {0-3, 12-15} of one
// k-quarter with rows {4-11} of the next, and NO row padding separates those (r01's 36-float rows: two 2-way conflicts
// per group, SQ_LDS_BANK_CONFLICT = 34 % of the kernel's cycles at M = 512); with the swizzle each group's sixteen
// 16-byte reads fall on sixteen different bank quads, and the eight lanes of a ds_write_b128 group fill one row.  Loads of block i+1 are in flight behind the MFMAs of block i (double-buffered LDS, one barrier
// per block).  chunks == 1: fused epilogue.  chunks > 1: the chunk's partial tiles go to a workspace slab in
// MFMA register layout and gemm_reduce_kernel sums the slabs left to right (canonical order) and runs the epilogue.
#define DSM_XS_LD 32
__device__ __forceinline__ int dsm_xs_sw(int row) { return ((row >> 1) & 1) | (((row >> 3) & 1) << 2); }
template <typename WT, typename KVT, int MT, int NT, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_tile_kernel(GemmArgs a) {
  __shared__ __attribute__((aligned(16))) float Xs[2][16 * MT][DSM_XS_LD];
  launch_stamp_begin(a.ts);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int xu0 = 4 * ((2 * q) ^ dsm_xs_sw(r)), xu1 = xu0 ^ 4;  // this lane's two 16-byte units of a fragment row (swizzled)
  const int chunks = gridDim.y;  // split-K across workgroups; 1 when a.chunk_loop walks the chunks below
  const int m_base = blockIdx.z * (16 * MT);
  const int n_base = blockIdx.x * 64 + 16 * wave;  // this wave's n-tile
  const WT* W = reinterpret_cast<const WT*>(a.W);

  const WT* wrow[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) wrow[nt] = dsm_wbase<WT>(a, W, n_base + nt * a.nt_stride, r, q);
  const int wst = dsm_wstep<WT>(a);
  // cooperative X fetch: 16*MT rows x 8 float4 pieces; thread t takes piece t (and t+256 when MT == 4).
  // Scalars, not arrays: small per-thread arrays ended up in scratch memory here.
  constexpr int PIECES = 16 * MT * 8;
  constexpr bool TWO = PIECES > 256;
  const bool has0 = tid < PIECES;
  const int row0 = has0 ? (tid >> 3) : 0, part = tid & 7;  // !has0: loads row 0 (valid), stores nothing
  int m0 = m_base + row0;
  m0 = m0 < a.M ? m0 : a.M - 1;
  const float* xsrc0 = a.X + a.xmap.off(m0) + 4 * part;
  const int xdst0 = row0 * DSM_XS_LD + 4 * (part ^ dsm_xs_sw(row0));
  int m1 = m_base + row0 + 32;
  m1 = m1 < a.M ? m1 : a.M - 1;
  const float* xsrc1 = a.X + a.xmap.off(m1) + 4 * part;
  const int xdst1 = (row0 + 32) * DSM_XS_LD + 4 * (part ^ dsm_xs_sw(row0));  // +32 leaves bits 1 and 3 of the row alone

  const int nloop = a.chunk_loop > 1 ? a.chunk_loop : 1;
  f32x4 acc[NT][MT], tot[NT][MT];
  float xbA[MT][8], xbB[MT][8];  // activation fragments: even blocks in A, odd blocks in B
  for (int cl = 0; cl < nloop; ++cl) {
  const int chunk = a.chunk_loop > 1 ? cl : (int)blockIdx.y;
  const int k0 = chunk * DSM_KC;
  const int k1 = min(k0 + DSM_KC, a.Kpad);
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // Every load of the chunk — 8 activation pieces and 8 weight fragments per thread — is issued up front, in the
  // order the blocks consume them and without a branch around any of them (block indices are clamped to the chunk's
  // last block; threads without an activation piece load a row they never store).  A load under a branch makes the
  // compiler wait for it on the spot, and s_waitcnt counts loads in issue order, so fetching "two blocks ahead"
  // inside the loop had been serialising every block behind the HBM latency: half the MFMA rate.  Now the latency
  // is paid once per workgroup, while the co-resident workgroups (3-4 per CU) keep the MFMA pipe busy.
  const int nkb = (k1 - k0) >> 5;  // 1..8 blocks of 32
  // named scalars, not arrays: the 8-entry prefetch arrays were left in scratch memory by the compiler
#define DSM_FOR8(F) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7)
#define DSM_LOADBLK(I)                                                                 \
  float4 xp##I, xq##I;                                                                 \
  Raw8<WT> rw##I[NT];                                                                  \
  {                                                                                    \
    const int kb_ = k0 + 32 * ((I) < nkb ? (I) : nkb - 1);                             \
    xp##I = *reinterpret_cast<const float4*>(xsrc0 + kb_);                             \
    xq##I = TWO ? *reinterpret_cast<const float4*>(xsrc1 + kb_) : xp##I;               \
    _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) rw##I[nt].load(wrow[nt] + (long)kb_ * wst);  \
  }
  DSM_FOR8(DSM_LOADBLK)
  __builtin_amdgcn_sched_barrier(0);  // all requests issued before the first block waits for its own
  // Fragments of block i+1 are read from LDS while the MFMAs of block i run (r02): a block opens with its first MFMA
  // group right behind the barrier instead of "store, barrier, wait for the LDS read".  One barrier per block: block
  // i+1's activations are stored before it and read after it; the buffer they overwrite was last read (as block i-1)
  // before the previous barrier.
#define DSM_XSTORE(I)                                                                  \
  {                                                                                    \
    float* xs_ = &Xs[(I) & 1][0][0];                                                   \
    if (has0) *reinterpret_cast<float4*>(xs_ + xdst0) = xp##I;                         \
    if (TWO) *reinterpret_cast<float4*>(xs_ + xdst1) = xq##I;                          \
  }
#define DSM_XFRAG(XB, I)                                                               \
  _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) {                                  \
    const float* fp = &Xs[(I) & 1][0][0] + (16 * mt + r) * DSM_XS_LD;                  \
    const float4 f0 = *reinterpret_cast<const float4*>(fp + xu0), f1 = *reinterpret_cast<const float4*>(fp + xu1); \
    XB[mt][0] = f0.x; XB[mt][1] = f0.y; XB[mt][2] = f0.z; XB[mt][3] = f0.w;            \
    XB[mt][4] = f1.x; XB[mt][5] = f1.y; XB[mt][6] = f1.z; XB[mt][7] = f1.w;            \
  }
#define DSM_XMFMA(CUR, S0, S1)                                                         \
  _Pragma("unroll") for (int s = (S0); s < (S1); ++s) {                                \
    _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                  \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                  \
      acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[nt][s], CUR[mt][s], acc[nt][mt], 0, 0, 0); \
    if (NT * MT > 1) __builtin_amdgcn_sched_barrier(0); /* round-robin over the accumulators */ \
  }
  // PIPE: the two-fragment-set form.  The gate's two n-tiles per wave do not have the registers for it (178 VGPRs: one
  // workgroup per CU less; 16x4 m-tiles: spills), so NT = 2 keeps the plain order: store, barrier, read, multiply.
  constexpr bool PIPE = NT == 1;
  if (PIPE) {
    if (cl > 0) __syncthreads();  // chunk loop: the previous chunk's last fragment reads are done before Xs[0] is rewritten
    DSM_XSTORE(0)
    __syncthreads();
    DSM_XFRAG(xbA, 0)
  }
#define DSM_BLOCK(I, IN, CUR, NXT)                                                     \
  if ((I) < nkb) { /* workgroup-uniform */                                             \
    float wa[NT][8];                                                                   \
    _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) rw##I[nt].unpack(wa[nt]);        \
    if (PIPE) {                                                                        \
      if ((I) + 1 < nkb) DSM_XSTORE(IN)                                                \
      __syncthreads();                                                                 \
      DSM_XMFMA(CUR, 0, 1)                                                             \
      if ((I) + 1 < nkb) { DSM_XFRAG(NXT, IN) }                                        \
      __builtin_amdgcn_sched_barrier(0);                                               \
      DSM_XMFMA(CUR, 1, 8)                                                             \
    } else {                                                                           \
      DSM_XSTORE(I)                                                                    \
      __syncthreads();                                                                 \
      DSM_XFRAG(xbA, I)                                                                \
      DSM_XMFMA(xbA, 0, 8)                                                             \
    }                                                                                  \
  }
  DSM_BLOCK(0, 1, xbA, xbB) DSM_BLOCK(1, 2, xbB, xbA) DSM_BLOCK(2, 3, xbA, xbB) DSM_BLOCK(3, 4, xbB, xbA)
  DSM_BLOCK(4, 5, xbA, xbB) DSM_BLOCK(5, 6, xbB, xbA) DSM_BLOCK(6, 7, xbA, xbB) DSM_BLOCK(7, 7, xbB, xbA)
#undef DSM_FOR8
#undef DSM_LOADBLK
#undef DSM_BLOCK
#undef DSM_XSTORE
#undef DSM_XFRAG
#undef DSM_XMFMA
  // canonical split-K order: chunk sums added left to right (here in registers, otherwise by the reduce kernels)
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) tot[nt][mt] = cl == 0 ? acc[nt][mt] : tot[nt][mt] + acc[nt][mt];
  }  // chunk loop
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = tot[nt][mt];

  if (chunks > 1) {
    // slab[chunk][m][n] f32, row-major with ld = ws_ntiles*16: a lane stores its 4 consecutive n of row m
    const long ld = (long)a.ws_ntiles * 16;
    const long mpad = (long)((a.M + 15) >> 4) * 16;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m = m_base + 16 * mt + r;
      if (m >= mpad) continue;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int n = n_base + nt * a.nt_stride + 4 * q;
        *reinterpret_cast<f32x4*>(a.ws + ((long)blockIdx.y * mpad + m) * ld + n) = acc[nt][mt];
      }
    }
    launch_stamp_end(a.ts);
    return;
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = m_base + 16 * mt + r;
    if (EPI == EPI_GATE) {
      epi_gate(a, acc[0][mt], acc[NT - 1][mt], m, n_base + 4 * q);
    } else {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int n = n_base + nt * a.nt_stride + 4 * q;
        if (EPI == EPI_RVQ)
          epi_rvq(a, acc[nt][mt], m, n, (n_base + nt * a.nt_stride) >> 4, q);
        else
          epi_store_qkv<KVT, EPI>(a, acc[nt][mt], m, n);
      }
    }
  }
  launch_stamp_end(a.ts);
}

// ---- tiled GEMM, whole K inside the workgroup (large batches: the (n, m) tiles alone fill the chip, so there is no
// split-K across workgroups, no slabs, no reduce launch).  Same tile, same LDS staging, same canonical order as
// gemm_tile_kernel — chunk sums of 256 k are formed in `acc` and added left to right into `tot` — but the loads run as a
// ROLLING window over the 32-wide blocks of the whole K range: four register slots hold blocks g .. g+3, and as soon
// as block g has been consumed its slot requests block g+4, across chunk boundaries.  (The r01 chunk loop inside
// gemm_tile_kernel requested a whole chunk only after finishing the previous one: every 8 blocks the MFMA pipe of the
// wave drained behind an L2 / HBM round trip — 64 % of the f32-MFMA rate at B >= 512.)
// Named scalars and a hand-unrolled body, not arrays: see gemm_tile_kernel.  Window depth D: 4 blocks, or 2 where a
// block is twice the MFMA work (the gate's two n-tiles per wave) or twice the registers (f32 weights) — the same ~4096
// MFMA cycles of cover, and the four-slot form of those two spilled.
template <typename WT, int NT>
struct LoopDepth {
  static constexpr int MAX = (NT == 2 || sizeof(WT) == 4) ? 2 : 4;
};
template <typename WT, typename KVT, int MT, int NT, int EPI, int D, int OCC = 2>
__global__ __launch_bounds__(256, OCC) void gemm_loop_kernel(GemmArgs a) {
  static_assert(D == 2 || D == 4, "window depth");
  launch_stamp_begin(a.ts);
  __shared__ __attribute__((aligned(16))) float Xs[2][16 * MT][DSM_XS_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int xu0 = 4 * ((2 * q) ^ dsm_xs_sw(r)), xu1 = xu0 ^ 4;  // this lane's two 16-byte units of a fragment row (swizzled)
  const int m_base = blockIdx.z * (16 * MT);
  const int n_base = blockIdx.x * 64 + 16 * wave;
  const WT* W = reinterpret_cast<const WT*>(a.W);
  const WT* wrow[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) wrow[nt] = dsm_wbase<WT>(a, W, n_base + nt * a.nt_stride, r, q);
  const int wst = dsm_wstep<WT>(a);
  constexpr int PIECES = 16 * MT * 8;
  constexpr bool TWO = PIECES > 256;
  const bool has0 = tid < PIECES;
  const int row0 = has0 ? (tid >> 3) : 0, part = tid & 7;
  int m0 = m_base + row0;
  m0 = m0 < a.M ? m0 : a.M - 1;
  const float* xsrc0 = a.X + a.xmap.off(m0) + 4 * part;
  const int xdst0 = row0 * DSM_XS_LD + 4 * (part ^ dsm_xs_sw(row0));
  int m1 = m_base + row0 + 32;
  m1 = m1 < a.M ? m1 : a.M - 1;
  const float* xsrc1 = a.X + a.xmap.off(m1) + 4 * part;
  const int xdst1 = (row0 + 32) * DSM_XS_LD + 4 * (part ^ dsm_xs_sw(row0));  // +32 leaves bits 1 and 3 of the row alone
  const int nb = a.Kpad >> 5;  // 32-wide blocks of the whole reduction

  f32x4 acc[NT][MT], tot[NT][MT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      acc[nt][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      tot[nt][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  float4 xp0, xp1, xp2, xp3, xq0, xq1, xq2, xq3;
  Raw8<WT> rw0[NT], rw1[NT], rw2[NT], rw3[NT];
  // block indices are clamped to the last block (loaded again, never used): no branch around any load
#define DSM_LLOAD(S, G)                                                              \
  {                                                                                  \
    const int kb_ = 32 * min((G), nb - 1);                                           \
    xp##S = *reinterpret_cast<const float4*>(xsrc0 + kb_);                           \
    xq##S = TWO ? *reinterpret_cast<const float4*>(xsrc1 + kb_) : xp##S;             \
    _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) rw##S[nt].load(wrow[nt] + (long)kb_ * wst); \
  }
  DSM_LLOAD(0, 0) DSM_LLOAD(1, 1)
  if (D == 4) { DSM_LLOAD(2, 2) DSM_LLOAD(3, 3) }
  __builtin_amdgcn_sched_barrier(0);
  // The activation fragments of block g+1 are fetched from LDS while the MFMAs of block g run (two register sets, A for even
  // blocks, B for odd ones): a block no longer opens with "store to LDS, barrier, wait for the LDS read" in front of its
  // first MFMA — with two waves per SIMD running the same program those gaps line up and idle the matrix pipe (PMC: 64 %
  // MFMA-busy at M = 512 before, profiles/r02).  One barrier per block remains; what it orders: block g+1's activations
  // are stored before it and read after it, and the buffer they overwrite was last read (as block g-1) before the
  // previous barrier.
  float xbA[MT][8], xbB[MT][8];
#define DSM_LSTORE(S, BUF)                                                           \
  {                                                                                  \
    float* xs_ = &Xs[BUF][0][0];                                                     \
    if (has0) *reinterpret_cast<float4*>(xs_ + xdst0) = xp##S;                       \
    if (TWO) *reinterpret_cast<float4*>(xs_ + xdst1) = xq##S;                        \
  }
#define DSM_LFRAG(XB, BUF)                                                           \
  _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) {                                \
    const float* fp = &Xs[BUF][0][0] + (16 * mt + r) * DSM_XS_LD;                    \
    const float4 f0 = *reinterpret_cast<const float4*>(fp + xu0), f1 = *reinterpret_cast<const float4*>(fp + xu1); \
    XB[mt][0] = f0.x; XB[mt][1] = f0.y; XB[mt][2] = f0.z; XB[mt][3] = f0.w;          \
    XB[mt][4] = f1.x; XB[mt][5] = f1.y; XB[mt][6] = f1.z; XB[mt][7] = f1.w;          \
  }
  DSM_LSTORE(0, 0)
  __syncthreads();
  DSM_LFRAG(xbA, 0)
  // S: this block's slot, SN: the next block's slot, CUR / NXT: fragment sets (g % D == 0 and D is even: block parity == S parity)
#define DSM_LMFMA(CUR, S0, S1)                                                       \
  _Pragma("unroll") for (int s = (S0); s < (S1); ++s) {                              \
    _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                \
      acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[nt][s], CUR[mt][s], acc[nt][mt], 0, 0, 0); \
    if (NT * MT > 1) __builtin_amdgcn_sched_barrier(0); /* round-robin over the accumulators */ \
  }
#define DSM_LSTEP(S, SN, CUR, NXT)                                                   \
  {                                                                                  \
    const int gb = g + (S);                                                          \
    float wa[NT][8];                                                                 \
    _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) rw##S[nt].unpack(wa[nt]);      \
    if (gb < nb) { /* workgroup-uniform */                                           \
      if (gb + 1 < nb) DSM_LSTORE(SN, ((S) + 1) & 1)                                 \
      __syncthreads();                                                               \
      /* the first MFMA group goes out right behind the barrier; the LDS reads of the NEXT block's fragments follow it   \
         (issued in front, the compiler makes every MFMA of this block wait for them) */ \
      DSM_LMFMA(CUR, 0, 1)                                                           \
      if (gb + 1 < nb) { DSM_LFRAG(NXT, ((S) + 1) & 1) }                             \
      __builtin_amdgcn_sched_barrier(0);                                             \
    }                                                                                \
    DSM_LLOAD(S, gb + D) /* the slot is free again: D blocks ahead, whatever chunk that is */ \
    __builtin_amdgcn_sched_barrier(0);                                               \
    if (gb < nb) {                                                                   \
      DSM_LMFMA(CUR, 1, 8)                                                           \
      if ((gb & 7) == 7 || gb == nb - 1) { /* a 256-wide chunk is complete: canonical left-to-right chunk sum */ \
        _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                            \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) {                          \
          tot[nt][mt] = tot[nt][mt] + acc[nt][mt]; /* first chunk: +0 + acc == acc bit for bit (acc starts at +0, so it is never -0) */ \
          acc[nt][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};                                 \
        }                                                                            \
      }                                                                              \
    }                                                                                \
  }
#pragma clang loop unroll(disable)
  for (int g = 0; g < nb; g += D) {
    if (D == 4) {
      DSM_LSTEP(0, 1, xbA, xbB) DSM_LSTEP(1, 2, xbB, xbA) DSM_LSTEP(2, 3, xbA, xbB) DSM_LSTEP(3, 0, xbB, xbA)
    } else {
      DSM_LSTEP(0, 1, xbA, xbB) DSM_LSTEP(1, 0, xbB, xbA)
    }
  }
#undef DSM_LSTORE
#undef DSM_LFRAG
#undef DSM_LMFMA
#undef DSM_LLOAD
#undef DSM_LSTEP
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = m_base + 16 * mt + r;
    if (EPI == EPI_GATE) {
      epi_gate(a, tot[0][mt], tot[NT - 1][mt], m, n_base + 4 * q);
    } else {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int n = n_base + nt * a.nt_stride + 4 * q;
        if (EPI == EPI_RVQ)
          epi_rvq(a, tot[nt][mt], m, n, (n_base + nt * a.nt_stride) >> 4, q);
        else
          epi_store_qkv<KVT, EPI>(a, tot[nt][mt], m, n);
      }
    }
  }
  launch_stamp_end(a.ts);
}

// ---- dot_mode 1 ("bx3"): the bf16-weight GEMMs on v_mfma_f32_16x16x32_bf16 (r03) -------------------------------------------
// Same tile, same grid, same slabs / epilogues as gemm_tile_kernel and gemm_loop_kernel — only the dot product of a K-chunk
// differs: per 32-wide block  acc = mfma(w, x_lo, acc); acc = mfma(w, x_mid, acc); acc = mfma(w, x_hi, acc)  with x = hi + mid +
// lo the exact three-bf16 split of dsm_split3 (every product exact; the instruction's adder: dsm_bf16_mfma_model.h, restated
// by the oracle's orc_linear_bx3).  3 x 16 matrix-pipe cycles per 32 k and tile instead of 8 x 32, and the matrix pipe no
// longer holds the vector ALU (experiments/fused_roles_probe.hip).  The activation block is split while it is staged: LDS holds
// three bf16 planes [16 MT rows][32 k] (64-byte rows, 16-byte units XOR-swizzled by bits 1-2 of the row), double-buffered, one
// barrier per block; weights go from HBM to the A operand as they are (eight bf16 = one 16-byte load per lane and block).
// LOOP: one workgroup walks every K-chunk (grid.y = 1) and adds the chunk sums left to right in registers; otherwise
// blockIdx.y is the chunk and, with more than one, the partial tile goes to the split-K slab.
template <typename KVT, int MT, int NT, int EPI, bool LOOP>
__global__ __launch_bounds__(256, 2) void gemm_bx3_kernel(GemmArgs a) {
  __shared__ __attribute__((aligned(16))) uint16_t Xp[2][3][16 * MT][32];
  launch_stamp_begin(a.ts);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int chunks = LOOP ? 1 : (int)gridDim.y;
  const int m_base = blockIdx.z * (16 * MT);
  const int n_base = blockIdx.x * (a.wg_cols ? a.wg_cols : 64) + 16 * wave;
  const uint16_t* W = reinterpret_cast<const uint16_t*>(a.W);
  const uint16_t* wrow[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) wrow[nt] = dsm_wbase<uint16_t>(a, W, n_base + nt * a.nt_stride, r, q);
  const int wst = dsm_wstep<uint16_t>(a);
  constexpr int PIECES = 16 * MT * 8;
  constexpr bool TWO = PIECES > 256;
  const bool has0 = tid < PIECES;
  const int row0 = has0 ? (tid >> 3) : 0, part = tid & 7;
  int m0 = m_base + row0;
  m0 = m0 < a.M ? m0 : a.M - 1;
  const float* xsrc0 = a.X + a.xmap.off(m0) + 4 * part;
  int m1 = m_base + row0 + 32;
  m1 = m1 < a.M ? m1 : a.M - 1;
  const float* xsrc1 = a.X + a.xmap.off(m1) + 4 * part;
  // four consecutive k of one row -> 8 bytes in each plane
  auto stage = [&](int buf, const float4& v, int row) {
    uint16_t h[4], m[4], l[4];
    dsm_split3(v.x, &h[0], &m[0], &l[0]); dsm_split3(v.y, &h[1], &m[1], &l[1]);
    dsm_split3(v.z, &h[2], &m[2], &l[2]); dsm_split3(v.w, &h[3], &m[3], &l[3]);
    const int off = row * 32 + (((part >> 1) ^ ((row >> 1) & 3)) * 8) + (part & 1) * 4;
    *reinterpret_cast<uint2*>(&Xp[buf][0][0][0] + off) = make_uint2((uint32_t)l[0] | ((uint32_t)l[1] << 16), (uint32_t)l[2] | ((uint32_t)l[3] << 16));
    *reinterpret_cast<uint2*>(&Xp[buf][1][0][0] + off) = make_uint2((uint32_t)m[0] | ((uint32_t)m[1] << 16), (uint32_t)m[2] | ((uint32_t)m[3] << 16));
    *reinterpret_cast<uint2*>(&Xp[buf][2][0][0] + off) = make_uint2((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16));
  };
  const int kb0 = LOOP ? 0 : (int)blockIdx.y * (DSM_KC >> 5);
  const int kb1 = LOOP ? (a.Kpad >> 5) : min(kb0 + (DSM_KC >> 5), a.Kpad >> 5);
  f32x4 acc[NT][MT], tot[NT][MT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      acc[nt][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      tot[nt][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  float4 xa = *reinterpret_cast<const float4*>(xsrc0 + 32 * kb0);
  float4 xb = TWO ? *reinterpret_cast<const float4*>(xsrc1 + 32 * kb0) : xa;
  uint4 wv[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) wv[nt] = *reinterpret_cast<const uint4*>(wrow[nt] + (long)32 * kb0 * wst);
  for (int g = kb0; g < kb1; ++g) {
    const int buf = (g - kb0) & 1;
    if (has0) stage(buf, xa, row0);
    if (TWO) stage(buf, xb, row0 + 32);
    dsm_bf16x8 wa[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const uint4 w = wv[nt];
      wa[nt][0] = (short)(w.x & 0xFFFF); wa[nt][1] = (short)(w.x >> 16); wa[nt][2] = (short)(w.y & 0xFFFF); wa[nt][3] = (short)(w.y >> 16);
      wa[nt][4] = (short)(w.z & 0xFFFF); wa[nt][5] = (short)(w.z >> 16); wa[nt][6] = (short)(w.w & 0xFFFF); wa[nt][7] = (short)(w.w >> 16);
    }
    {  // the next block's loads fly behind this block's MFMAs (clamped to the last block: loaded again, never used)
      const int gn = g + 1 < kb1 ? g + 1 : g;
      xa = *reinterpret_cast<const float4*>(xsrc0 + 32 * gn);
      if (TWO) xb = *reinterpret_cast<const float4*>(xsrc1 + 32 * gn);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) wv[nt] = *reinterpret_cast<const uint4*>(wrow[nt] + (long)32 * gn * wst);
    }
    __syncthreads();  // block g's planes are complete; buffer buf ^ 1 was last read before the previous barrier
#pragma unroll
    for (int p = 0; p < 3; ++p)  // canonical order of the pieces: lo, mid, hi
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int row = 16 * mt + r;
        const dsm_bf16x8 xf = *reinterpret_cast<const dsm_bf16x8*>(&Xp[buf][p][0][0] + row * 32 + ((q ^ ((row >> 1) & 3)) * 8));
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[nt], xf, acc[nt][mt], 0, 0, 0);
      }
    if (LOOP && ((g & 7) == 7 || g == kb1 - 1)) {  // a 256-wide chunk is complete: canonical left-to-right chunk sum
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          tot[nt][mt] = tot[nt][mt] + acc[nt][mt];  // first chunk: +0 + acc (the oracle does the same)
          acc[nt][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }
  }
  if (!LOOP) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) tot[nt][mt] = (f32x4){0.f, 0.f, 0.f, 0.f} + acc[nt][mt];  // +0 + chunk sum, as above
  }
  if (chunks > 1) {
    const long ld = (long)a.ws_ntiles * 16;
    const long mpad = (long)((a.M + 15) >> 4) * 16;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m = m_base + 16 * mt + r;
      if (m >= mpad) continue;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int n = n_base + nt * a.nt_stride + 4 * q;
        *reinterpret_cast<f32x4*>(a.ws + ((long)blockIdx.y * mpad + m) * ld + n) = tot[nt][mt];
      }
    }
    launch_stamp_end(a.ts);
    return;
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = m_base + 16 * mt + r;
    if (EPI == EPI_GATE) {
      epi_gate(a, tot[0][mt], tot[NT - 1][mt], m, n_base + 4 * q);
    } else {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) epi_store_qkv<KVT, EPI>(a, tot[nt][mt], m, n_base + nt * a.nt_stride + 4 * q);
    }
  }
  launch_stamp_end(a.ts);
}

// Ordered sum of `chunks` slab values at p, p + cstride, ...: loads are issued eight at a time, the adds stay
// strictly left to right (canonical split-K order).
__device__ __forceinline__ f32x4 slab_sum(const float* p, long cstride, int chunks) {
  f32x4 t = *reinterpret_cast<const f32x4*>(p);
  for (int c0 = 1; c0 < chunks; c0 += 8) {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int c = c0 + u < chunks ? c0 + u : chunks - 1;  // clamped duplicate load, discarded below
      v[u] = *reinterpret_cast<const f32x4*>(p + c * cstride);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (c0 + u < chunks) t = t + v[u];
  }
  return t;
}

// Ordered split-K reduce + epilogue: one wave per 16x16 output tile (a gate/up tile pair for EPI_GATE).
template <typename KVT, int EPI>
__global__ __launch_bounds__(256) void gemm_reduce_kernel(GemmArgs a, int chunks) {
  launch_stamp_begin(a.ts ? a.ts + 2 : nullptr);  // the reduce launch owns the record after its GEMM's
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int mtiles = (a.M + 15) >> 4;
  const int out_ntiles = (a.N + 15) >> 4;  // gate: a.N is the hidden width
  const int t = blockIdx.x * 4 + wave;
  if (t >= mtiles * out_ntiles) return;
  const int mtile = t / out_ntiles, ntile = t % out_ntiles;
  constexpr int NT = (EPI == EPI_GATE) ? 2 : 1;
  const long ld = (long)a.ws_ntiles * 16, cstride = (long)mtiles * 16 * ld;
  const int m = mtile * 16 + r, n = ntile * 16 + 4 * q;
  f32x4 tot[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    tot[nt] = slab_sum(a.ws + (long)m * ld + n + nt * a.nt_stride, cstride, chunks);
  }
  if (EPI == EPI_GATE)
    epi_gate(a, tot[0], tot[NT - 1], m, n);
  else if (EPI == EPI_RVQ)
    epi_rvq(a, tot[0], m, n, ntile, q);
  else
    epi_store_qkv<KVT, EPI>(a, tot[0], m, n);
  launch_stamp_end(a.ts ? a.ts + 2 : nullptr);
}

#include "dsm_gemm_wk.h"  // whole-K-in-the-workgroup GEMM for M <= 64 (r04): no slabs, no reduce launch

// Canonical row reduction (dsm_numerics.h): 256 threads per row, thread t owns elements 1024*it + 4*t + j;
// wave butterflies, then the 4 wave totals left to right.  `red` = 8 floats of LDS.
__device__ __forceinline__ void block_row_sums(float& s, float& s2, float* red) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  s = wave_sum64(s);
  s2 = wave_sum64(s2);
  if (lane == 0) { red[wave] = s; red[4 + wave] = s2; }
  __syncthreads();
  s = ((red[0] + red[1]) + red[2]) + red[3];
  s2 = ((red[4] + red[5]) + red[6]) + red[7];
}

#define DSM_ROW_ITS 4  // d_model <= 4096

__device__ __forceinline__ void row_norm_apply(const float4 (&v)[DSM_ROW_ITS], float s, float s2, int d, float eps, int rms,
                                               const float* __restrict__ w, const float* __restrict__ b,
                                               float* __restrict__ out) {
  float mm = 1.f, mean = 0.f, inv = 0.f;
  if (rms) {
    mm = sqrtf(s2 / (float)d + eps);
  } else {
    mean = s / (float)d;
    float var = s2 / (float)d - mean * mean;
    inv = 1.0f / sqrtf(var + eps);
  }
#pragma unroll
  for (int it = 0; it < DSM_ROW_ITS; ++it) {
    const int i = it * 1024 + 4 * (int)threadIdx.x;
    if (i < d) {
      const float4 wv = *reinterpret_cast<const float4*>(w + i);
      float4 o;
      if (rms) {
        o.x = (v[it].x / mm) * wv.x; o.y = (v[it].y / mm) * wv.y; o.z = (v[it].z / mm) * wv.z; o.w = (v[it].w / mm) * wv.w;
      } else {
        const float4 bv = *reinterpret_cast<const float4*>(b + i);
        o.x = ((v[it].x - mean) * inv) * wv.x + bv.x; o.y = ((v[it].y - mean) * inv) * wv.y + bv.y;
        o.z = ((v[it].z - mean) * inv) * wv.z + bv.z; o.w = ((v[it].w - mean) * inv) * wv.w + bv.w;
      }
      *reinterpret_cast<float4*>(out + i) = o;
    }
  }
}

// Ordered split-K reduce + STORE epilogue (bias / act / scale / residual) + the row norm that follows it, one
// workgroup per output row (N = d_model).  Same per-element arithmetic as epi_store_qkv<EPI_STORE> followed by
// row_norm_kernel, and the same canonical row reduction (dsm_numerics.h: thread t of 256 chains its elements
// 1024*it + 4*t + j over it = 0, 1, ...) — but the work of "thread t" is spread over ITS threads, one per 1024-element group
// (workgroup of 256 * ITS threads), so that every slab load of the row is in flight at once: a launch has only M
// workgroups (32 per stream group at B = 64) and its time is the number of dependent load rounds.  The chain over `it`
// is handed from group to group through LDS (ITS - 1 barriers of ALU work).  r01 form: 256 threads, eight chunks of every
// group per round = three rounds for the 22-chunk ff_out reduce, 6.8 us per launch.
template <int ITS>  // 1024-element groups per row: ceil(d / 1024) rounded up to 1, 2 or 4
__global__ __launch_bounds__(256 * ITS) void gemm_reduce_rows_kernel(GemmArgs a, int chunks) {
  constexpr int RS = ITS == 4 ? 16 : 24;  // chunk loads per round (VGPR budget: 1024 threads leave 128 each)
  __shared__ float red[8];
  __shared__ float carry[2][256];
  launch_stamp_begin(a.ts ? a.ts + 2 : nullptr);
  const int m = blockIdx.x;
  const int d = a.N;
  const int it = threadIdx.x >> 8, t = threadIdx.x & 255;
  const long ld = (long)a.ws_ntiles * 16, cstride = (long)((a.M + 15) >> 4) * 16 * ld;
  const int i = it * 1024 + 4 * t;
  const bool live = i < d;
  const float* p = a.ws + (long)m * ld + (live ? i : 4 * t);  // a group beyond d re-reads the row's first group and is dropped
  // The epilogue's operands — bias, LayerScale, residual, norm weights — do not depend on the sums: requested FIRST, unconditionally
  // (an absent operand re-reads the slab word, a valid address, and is never used), so that they travel with the slab loads instead
  // of costing two more dependent memory round trips behind them (r04: 5.6 -> 4.x us per launch, 64 launches per step on the chain).
  const int ic = live ? i : 4 * t;
  const f32x4 bias4 = *reinterpret_cast<const f32x4*>(a.bias ? a.bias + ic : p);
  const f32x4 scale4 = *reinterpret_cast<const f32x4*>(a.scale ? a.scale + ic : p);
  const f32x4 res4 = *reinterpret_cast<const f32x4*>(a.res ? a.res + a.rmap.off(m) + ic : p);
  const f32x4 nw4 = *reinterpret_cast<const f32x4*>(a.norm_w + ic);
  const f32x4 nb4 = *reinterpret_cast<const f32x4*>(a.norm_rms ? p : a.norm_b + ic);
  f32x4 tot = *reinterpret_cast<const f32x4*>(p);
  for (int c0 = 1; c0 < chunks; c0 += RS) {
    f32x4 w[RS];
#pragma unroll
    for (int u = 0; u < RS; ++u) {
      const int c = c0 + u < chunks ? c0 + u : chunks - 1;  // clamped duplicate load, discarded below
      w[u] = *reinterpret_cast<const f32x4*>(p + c * cstride);
    }
    __builtin_amdgcn_sched_barrier(0);  // every load of the round is issued before the first add waits
#pragma unroll
    for (int u = 0; u < RS; ++u)
      if (c0 + u < chunks) tot = tot + w[u];
  }
  float o[4] = {tot[0], tot[1], tot[2], tot[3]};
  if (live) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (a.bias) o[j] = o[j] + bias4[j];
      if (a.act == 1) o[j] = dsm_gelu_erf(o[j]);
      if (a.scale) o[j] = o[j] * scale4[j];
    }
    if (a.res) {
      o[0] = res4[0] + o[0]; o[1] = res4[1] + o[1]; o[2] = res4[2] + o[2]; o[3] = res4[3] + o[3];
    }
    *reinterpret_cast<float4*>(a.Y + a.ymap.off(m) + i) = make_float4(o[0], o[1], o[2], o[3]);
  }
  // thread t's chain, continued group by group
  float s = 0.0f, s2 = 0.0f;
#pragma unroll
  for (int k = 0; k < ITS; ++k) {
    if (it == k) {
      if (k > 0) { s = carry[0][t]; s2 = carry[1][t]; }
      if (live) {
        s = s + o[0]; s2 = DSM_FMAF(o[0], o[0], s2);
        s = s + o[1]; s2 = DSM_FMAF(o[1], o[1], s2);
        s = s + o[2]; s2 = DSM_FMAF(o[2], o[2], s2);
        s = s + o[3]; s2 = DSM_FMAF(o[3], o[3], s2);
      }
      if (k + 1 < ITS) { carry[0][t] = s; carry[1][t] = s2; }
    }
    if (k + 1 < ITS) __syncthreads();
  }
  if (it == ITS - 1) {  // these 4 waves hold the 256 thread totals: wave butterflies, then the wave totals left to right
    const int lane = t & 63, wave = t >> 6;
    s = wave_sum64(s);
    s2 = wave_sum64(s2);
    if (lane == 0) { red[wave] = s; red[4 + wave] = s2; }
  }
  __syncthreads();
  s = ((red[0] + red[1]) + red[2]) + red[3];
  s2 = ((red[4] + red[5]) + red[6]) + red[7];
  if (live) {
    float mm = 1.f, mean = 0.f, inv = 0.f;
    if (a.norm_rms) {
      mm = sqrtf(s2 / (float)d + a.norm_eps);
    } else {
      mean = s / (float)d;
      float var = s2 / (float)d - mean * mean;
      inv = 1.0f / sqrtf(var + a.norm_eps);
    }
    float4 r;
    if (a.norm_rms) {
      r.x = (o[0] / mm) * nw4[0]; r.y = (o[1] / mm) * nw4[1]; r.z = (o[2] / mm) * nw4[2]; r.w = (o[3] / mm) * nw4[3];
    } else {
      r.x = ((o[0] - mean) * inv) * nw4[0] + nb4[0]; r.y = ((o[1] - mean) * inv) * nw4[1] + nb4[1];
      r.z = ((o[2] - mean) * inv) * nw4[2] + nb4[2]; r.w = ((o[3] - mean) * inv) * nw4[3] + nb4[3];
    }
    *reinterpret_cast<float4*>(a.norm_out + (long)m * d + i) = r;
  }
  launch_stamp_end(a.ts ? a.ts + 2 : nullptr);
}

// ------------------------------------------------------------------------------------------
// Row norms — one workgroup per row.  rms 1: RmsNorm (core/batched_transformer.rs:194-198),
// rms 0: LayerNorm (core/batched_transformer.rs:200-222).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void row_norm_kernel(float* __restrict__ y, const float* __restrict__ x,
                                                       const float* __restrict__ w, const float* __restrict__ b, int rows,
                                                       int d, float eps, int rms) {
  __shared__ float red[8];
  const int row = blockIdx.x;
  const float* xr = x + (long)row * d;
  float4 v[DSM_ROW_ITS];
  float s = 0.0f, s2 = 0.0f;
#pragma unroll
  for (int it = 0; it < DSM_ROW_ITS; ++it) {
    const int i = it * 1024 + 4 * (int)threadIdx.x;
    v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < d) {  // d % 4 == 0: a thread's four elements are in range together
      v[it] = *reinterpret_cast<const float4*>(xr + i);
      s = s + v[it].x; s2 = DSM_FMAF(v[it].x, v[it].x, s2);
      s = s + v[it].y; s2 = DSM_FMAF(v[it].y, v[it].y, s2);
      s = s + v[it].z; s2 = DSM_FMAF(v[it].z, v[it].z, s2);
      s = s + v[it].w; s2 = DSM_FMAF(v[it].w, v[it].w, s2);
    }
  }
  block_row_sums(s, s2, red);
  row_norm_apply(v, s, s2, d, eps, rms, w, b, y + (long)row * d);
}

// ------------------------------------------------------------------------------------------
// Attention over the ring cache — core/batched_transformer.rs:107-113 + mask semantics of
// core/kv_cache.rs:119-237 in closed form: slot j is visible to query t iff
//   j <= e1  and  (e1 - j) mod ctx >= T-1-t,   e1 = start_pos + T - 1.
// One workgroup (4 waves) per (slot b, head h).  LPK = HD/8 lanes per key, G = 64/LPK keys per
// wave-instruction, key j -> wave (j/G)%4, lane group j%G (canonical order, dsm_numerics.h).
// ------------------------------------------------------------------------------------------
// T = 1 only: the QKV GEMM left its split-K slabs in place and this kernel finishes the job for its own (slot, head) —
// ordered slab sums of the 3 x HD values, RoPE on q and k, K/V rows rounded into the ring cache at widx — before it
// attends.  Same arithmetic, same order as gemm_reduce_kernel<KVT, EPI_QKV>; one launch and the q round trip less.
struct AttnFused {
  const float* ws;  // null: q comes from qbuf, K/V are already in the cache
  long ld, cstride;
  int chunks;
  const float* rope_cs;
  const uint32_t* widx;
  int nt;  // K / V rows with non-temporal loads (DSM_ATTN_NT)
  int q_only;  // the slabs hold a [M][d] query projection only (cross-attention, r04): no K / V rows, no RoPE, nothing scattered
};

// UNRB: keys per lane group and batch on a bf16 ring (8: 126 VGPRs, 4: 78).  The key -> (wave, lane group) assignment and the order
// inside a group do not depend on it (same bits).  4 is faster outright at head_dim 64 (stt-2.6b: 39.2 against 42.8 us per 64-slot
// launch, 9.18 against 9.63 ms per step); at head_dim 128 the kernel itself is 3-4 % faster with 8 (launch_attn_t picks; r04, late;
// profiles/r04/experiments/timing_notes.txt).
template <typename KVT, int HD, int T, int UNRB = 8>
__global__ __launch_bounds__(256, 4) void attn_kernel(float* __restrict__ out, const float* __restrict__ qbuf,
                                                   const KVT* __restrict__ kcache, const KVT* __restrict__ vcache,
                                                   const uint32_t* __restrict__ start_pos,
                                                   const uint8_t* __restrict__ active, int H, int ctx, int d,
                                                   unsigned long long* __restrict__ ts, AttnFused fq) {
  constexpr int NW = 4, LPK = HD / 8, G = 64 / LPK;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  // optional device-clock bracket of the whole launch (dsm_prof_read_device): first workgroup in, last one out.
  // HIP events around a launch also count the time it queues behind other streams' kernels; this does not.
  // Only the first 256 workgroups race for the start stamp and the last 1024 for the end stamp (dispatch is in index
  // order): thousands of atomics on one address would themselves stretch a large launch.
  const bool ts_first = ts && threadIdx.x == 0 && blockIdx.x < 256;
  const bool ts_last = ts && threadIdx.x == 0 && blockIdx.x + 1024 >= gridDim.x;
  if (ts_first) atomicMin(&ts[0], wall_clock64());
  if (!active[b]) {  // inactive slots: output unused by the reference (core/asr.rs:221-223)
    if (ts_last) atomicMax(&ts[1], wall_clock64());
    return;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane / LPK, li = lane % LPK;
  float* sc = lds;                       // [T][ctx]
  float* red = lds + T * ctx;            // [NW][T][HD] then small scratch
  float* scratch = red + NW * T * HD;    // [16]

  const uint32_t sp = start_pos[b];
  const long e1 = (long)sp + T - 1;
  const int nvalid = (int)((e1 + 1 < (long)ctx) ? e1 + 1 : ctx);  // slots > e1 were never written
  const float scale = (float)(1.0 / sqrt((double)HD));

  const KVT* Kb = kcache + ((long)b * H + h) * ctx * HD;
  const KVT* Vb = vcache + ((long)b * H + h) * ctx * HD;
  if (T == 1 && fq.ws) {
    // row m = b of the [M][3d] QKV product: q at n = h*HD + i, k at d + ..., v at 2d + ... (core/batched_transformer.rs:77-82)
    if (tid < (fq.q_only ? HD / 4 : 3 * HD / 4)) {
      const int part = tid / (HD / 4), i0 = 4 * (tid % (HD / 4));
      const f32x4 v = slab_sum(fq.ws + (long)b * fq.ld + part * d + h * HD + i0, fq.cstride, fq.chunks);
      float o[4] = {v[0], v[1], v[2], v[3]};
      if (part < 2 && fq.rope_cs) {  // rope_i on interleaved pairs — core/transformer.rs:373-377
        const float4 cs = *reinterpret_cast<const float4*>(fq.rope_cs + ((long)b * (HD / 2) + (i0 >> 1)) * 2);
        const float co[2] = {cs.x, cs.z}, si[2] = {cs.y, cs.w};
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          float x0 = v[2 * p], x1 = v[2 * p + 1];
          float t0 = x0 * co[p], t1 = x1 * si[p], t2 = x0 * si[p], t3 = x1 * co[p];
          o[2 * p] = t0 - t1;
          o[2 * p + 1] = t2 + t3;
        }
      }
      if (part == 0) {
        red[i0] = o[0]; red[i0 + 1] = o[1]; red[i0 + 2] = o[2]; red[i0 + 3] = o[3];  // q staged in the (still unused) red area; ctx*4 B is not a multiple of 16: scalar accesses
      } else {
        KVT* row = const_cast<KVT*>(part == 1 ? Kb : Vb) + (long)fq.widx[b] * HD + i0;
        store_kv4(row, o);
      }
    }
    __syncthreads();  // q visible in LDS; this workgroup's K/V row visible to its own loads below
  }
  float qv[T][8];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    if (T == 1 && fq.ws) {
#pragma unroll
      for (int dd = 0; dd < 8; ++dd) qv[t][dd] = red[8 * li + dd];
    } else {
      const float* qp = qbuf + ((long)(b * T + t)) * d + h * HD + 8 * li;
      float4 q0 = *reinterpret_cast<const float4*>(qp), q1 = *reinterpret_cast<const float4*>(qp + 4);
      qv[t][0] = q0.x; qv[t][1] = q0.y; qv[t][2] = q0.z; qv[t][3] = q0.w;
      qv[t][4] = q1.x; qv[t][5] = q1.y; qv[t][6] = q1.z; qv[t][7] = q1.w;
    }
  }
  if (T == 1 && fq.ws) __syncthreads();  // every lane holds its q before `red` is reused for the partial outputs

  // ---- phase 1: scores.  Software-pipelined: while one batch of UNR keys per lane is reduced, the next batch's
  // 16-byte loads are already in flight (raw cache words, converted at use), so a wave keeps UNR..2*UNR loads
  // outstanding instead of draining to zero every iteration.  Out-of-range keys are clamped to the last valid row
  // (loaded, never used) so that the loads need no branches. ----
  constexpr int UNR = sizeof(KVT) == 2 ? UNRB : 4;
  constexpr int STEP = UNR * NW * G;
  const int jlast = nvalid - 1;
  const int last_slot = (int)(e1 % ctx);  // ring slot holding the newest key
  Raw8<KVT> ra[UNR], rb[UNR];
#define DSM_ISSUE(R, BASE, J0)                                                        \
  _Pragma("unroll") for (int u = 0; u < UNR; ++u) {                                    \
    const int jj = min((J0) + u * NW * G + g, jlast);                                  \
    if (fq.nt) R[u].load_nt(BASE + (long)jj * HD + 8 * li);                            \
    else R[u].load(BASE + (long)jj * HD + 8 * li);                                     \
  }
#define DSM_SCORES(R, J0)                                                              \
  _Pragma("unroll") for (int u = 0; u < UNR; ++u) {                                    \
    const int j = (J0) + u * NW * G + g;                                               \
    int delta = last_slot - min(j, jlast); /* == (e1 - j) mod ctx, no 64-bit division */ \
    delta += (delta < 0) ? ctx : 0;                                                    \
    float kv[8];                                                                       \
    R[u].unpack(kv);                                                                   \
    _Pragma("unroll") for (int t = 0; t < T; ++t) {                                    \
      float p = 0.0f;                                                                  \
      _Pragma("unroll") for (int dd = 0; dd < 8; ++dd) p = DSM_FMAF(qv[t][dd], kv[dd], p); \
      _Pragma("unroll") for (int off = LPK / 2; off >= 1; off >>= 1) p = p + __shfl_xor(p, off, 64); \
      if (li == 0 && j < nvalid) sc[t * ctx + j] = (delta >= T - 1 - t) ? p * scale : -DSM_INF_F; \
    }                                                                                  \
  }
  {
    int j0 = wave * G;
    DSM_ISSUE(ra, Kb, j0)
    while (j0 < nvalid) {
      DSM_ISSUE(rb, Kb, j0 + STEP)
      DSM_SCORES(ra, j0)
      j0 += STEP;
      if (j0 >= nvalid) break;
      DSM_ISSUE(ra, Kb, j0 + STEP)
      DSM_SCORES(rb, j0)
      j0 += STEP;
    }
  }
  // the first batch of V rows does not depend on the softmax: put it in flight across the statistics phase
  DSM_ISSUE(ra, Vb, wave * G)
  __syncthreads();

  // ---- phase 1.5: softmax statistics (softmax_last_dim) ----
  float lsum[T];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    float m = -DSM_INF_F;
    for (int j = tid; j < nvalid; j += 256) m = fmaxf(m, sc[t * ctx + j]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if (lane == 0) scratch[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(scratch[0], scratch[1]), fmaxf(scratch[2], scratch[3]));
    __syncthreads();
    float tp = 0.0f;
    for (int j = tid; j < nvalid; j += 256) {
      float p = dsm_expf(sc[t * ctx + j] - m);
      sc[t * ctx + j] = p;
      tp = tp + p;
    }
    tp = wave_sum64(tp);
    if (lane == 0) scratch[4 + wave] = tp;
    __syncthreads();
    lsum[t] = ((scratch[4] + scratch[5]) + scratch[6]) + scratch[7];
    __syncthreads();
  }

  // ---- phase 2: P.V ----
  float acc[T][8];
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int dd = 0; dd < 8; ++dd) acc[t][dd] = 0.0f;
#define DSM_PV(R, J0)                                                                  \
  _Pragma("unroll") for (int u = 0; u < UNR; ++u) { /* ascending j per (wave, group): canonical order */ \
    const int j = (J0) + u * NW * G + g;                                               \
    float vv[8];                                                                       \
    R[u].unpack(vv);                                                                   \
    _Pragma("unroll") for (int t = 0; t < T; ++t) {                                    \
      /* out-of-range keys get weight 0: fmaf(0, v, acc) == acc for the finite v of the clamped row */ \
      const float wgt = (j < nvalid) ? sc[t * ctx + j] / lsum[t] : 0.0f;               \
      _Pragma("unroll") for (int dd = 0; dd < 8; ++dd) acc[t][dd] = DSM_FMAF(wgt, vv[dd], acc[t][dd]); \
    }                                                                                  \
  }
  {
    int j0 = wave * G;  // ra already holds this batch
    while (j0 < nvalid) {
      DSM_ISSUE(rb, Vb, j0 + STEP)
      DSM_PV(ra, j0)
      j0 += STEP;
      if (j0 >= nvalid) break;
      DSM_ISSUE(ra, Vb, j0 + STEP)
      DSM_PV(rb, j0)
      j0 += STEP;
    }
  }
#undef DSM_ISSUE
#undef DSM_SCORES
#undef DSM_PV
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int dd = 0; dd < 8; ++dd) {
      float v = acc[t][dd];
#pragma unroll
      for (int off = 32; off >= LPK; off >>= 1) v = v + __shfl_xor(v, off, 64);
      acc[t][dd] = v;
    }
  if (g == 0) {
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int dd = 0; dd < 8; ++dd) red[(wave * T + t) * HD + 8 * li + dd] = acc[t][dd];
  }
  __syncthreads();
  for (int i = tid; i < T * HD; i += 256) {
    int t = i / HD, dd = i % HD;
    float tot = ((red[(0 * T + t) * HD + dd] + red[(1 * T + t) * HD + dd]) + red[(2 * T + t) * HD + dd]) +
                red[(3 * T + t) * HD + dd];
    out[((long)(b * T + t)) * d + h * HD + dd] = tot;
  }
  if (ts_last) atomicMax(&ts[1], wall_clock64());
}

// ---- attention over a SHORT ring (r04): T = 1, at most 32 positions, head_dim 64 — the DepFormer's rings (one position per
// codebook slice) and their cousins in small test models.  attn_kernel spends a 256-thread workgroup, five barriers and a
// 128-key software pipeline on what is at most 32 keys; here one WAVE owns a (slot, head) and a workgroup four of them.
// Same arithmetic, same canonical order: key j belongs to canonical wave j / 8 and lane group j % 8, so the physical wave walks
// the four canonical waves' keys (one key per lane group each), keeps the four partial outputs apart, folds each over its lane
// groups with the same butterfly (offsets 32, 16, 8) and adds them left to right; the softmax sum sits one probability per lane
// (thread j of canonical wave 0 in attn_kernel) under the same 64-lane butterfly, the other canonical waves contribute +0.
// Keys beyond the valid range contribute fma(0, v, +0) = +0 there and nothing here.
template <typename KVT, int HD>
__global__ __launch_bounds__(256) void attn_small_kernel(float* __restrict__ out, const float* __restrict__ qbuf,
                                                         const KVT* __restrict__ kcache, const KVT* __restrict__ vcache,
                                                         const uint32_t* __restrict__ start_pos, const uint8_t* __restrict__ active,
                                                         int BH, int H, int ctx, int d, AttnFused fq) {
  static_assert(HD == 64, "one lane group of 8 lanes per key");
  constexpr int LPK = HD / 8;
  __shared__ __attribute__((aligned(16))) float lds_all[4][32 + HD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int unit = blockIdx.x * 4 + wave;
  const int uc = unit < BH ? unit : BH - 1;
  const int b = uc / H, h = uc % H;
  const bool on = unit < BH && active[b];
  float* sc = lds_all[wave];  // [32] scores, then probabilities
  float* qs = sc + 32;        // [HD] the query (fused prologue)
  const int g = lane / LPK, li = lane % LPK;
  const uint32_t sp = start_pos[b];
  const long e1 = (long)sp;
  const int nvalid = (int)((e1 + 1 < (long)ctx) ? e1 + 1 : ctx);
  const float scale = (float)(1.0 / sqrt((double)HD));
  const KVT* Kb = kcache + ((long)b * H + h) * ctx * HD;
  const KVT* Vb = vcache + ((long)b * H + h) * ctx * HD;
  if (fq.ws) {  // attn_kernel's prologue, lane for thread
    if (on && lane < (fq.q_only ? HD / 4 : 3 * HD / 4)) {
      const int part = lane / (HD / 4), i0 = 4 * (lane % (HD / 4));
      const f32x4 v = slab_sum(fq.ws + (long)b * fq.ld + part * d + h * HD + i0, fq.cstride, fq.chunks);
      float o[4] = {v[0], v[1], v[2], v[3]};
      if (part < 2 && fq.rope_cs) {
        const float4 cs = *reinterpret_cast<const float4*>(fq.rope_cs + ((long)b * (HD / 2) + (i0 >> 1)) * 2);
        const float co[2] = {cs.x, cs.z}, si[2] = {cs.y, cs.w};
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          float x0 = v[2 * p], x1 = v[2 * p + 1];
          float t0 = x0 * co[p], t1 = x1 * si[p], t2 = x0 * si[p], t3 = x1 * co[p];
          o[2 * p] = t0 - t1;
          o[2 * p + 1] = t2 + t3;
        }
      }
      if (part == 0) {
        qs[i0] = o[0]; qs[i0 + 1] = o[1]; qs[i0 + 2] = o[2]; qs[i0 + 3] = o[3];
      } else {
        KVT* row = const_cast<KVT*>(part == 1 ? Kb : Vb) + (long)fq.widx[b] * HD + i0;
        store_kv4(row, o);
      }
    }
    __syncthreads();  // the new K / V row is visible to this workgroup's loads (every wave arrives: no wave has left yet)
  }
  if (!on) return;
  float qv[8];
  if (fq.ws) {
#pragma unroll
    for (int dd = 0; dd < 8; ++dd) qv[dd] = qs[8 * li + dd];
  } else {
    const float* qp = qbuf + (long)b * d + h * HD + 8 * li;
    const float4 q0 = *reinterpret_cast<const float4*>(qp), q1 = *reinterpret_cast<const float4*>(qp + 4);
    qv[0] = q0.x; qv[1] = q0.y; qv[2] = q0.z; qv[3] = q0.w; qv[4] = q1.x; qv[5] = q1.y; qv[6] = q1.z; qv[7] = q1.w;
  }
  const int jlast = nvalid - 1;
  Raw8<KVT> rk[4], rv[4];
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const int jj = min(8 * w + g, jlast);  // clamped: loaded, never used
    rk[w].load(Kb + (long)jj * HD + 8 * li);
  }
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const int jj = min(8 * w + g, jlast);
    rv[w].load(Vb + (long)jj * HD + 8 * li);
  }
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const int j = 8 * w + g;
    float kv[8];
    rk[w].unpack(kv);
    float p = 0.0f;
#pragma unroll
    for (int dd = 0; dd < 8; ++dd) p = DSM_FMAF(qv[dd], kv[dd], p);
#pragma unroll
    for (int off = LPK / 2; off >= 1; off >>= 1) p = p + __shfl_xor(p, off, 64);
    if (li == 0 && j < nvalid) sc[j] = p * scale;  // T = 1: every written slot is visible
  }
  // softmax_last_dim: thread j of canonical wave 0 owns probability j
  float m = -DSM_INF_F;
  if (lane < nvalid) m = fmaxf(m, sc[lane]);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  float tp = 0.0f;
  if (lane < nvalid) {
    const float p = dsm_expf(sc[lane] - m);
    sc[lane] = p;
    tp = tp + p;
  }
  tp = wave_sum64(tp);
  const float lsum = ((tp + 0.0f) + 0.0f) + 0.0f;  // canonical waves 1..3 hold no probability
  float tot[8];
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const int j = 8 * w + g;
    float vv[8];
    rv[w].unpack(vv);
    const float wgt = (j < nvalid) ? sc[j] / lsum : 0.0f;
#pragma unroll
    for (int dd = 0; dd < 8; ++dd) {
      float v = (j < nvalid) ? DSM_FMAF(wgt, vv[dd], 0.0f) : 0.0f;
#pragma unroll
      for (int off = 32; off >= LPK; off >>= 1) v = v + __shfl_xor(v, off, 64);
      tot[dd] = w == 0 ? v : tot[dd] + v;  // ((r0 + r1) + r2) + r3
    }
  }
  if (g == 0) {
    float* o = out + (long)b * d + h * HD + 8 * li;
    *reinterpret_cast<float4*>(o) = make_float4(tot[0], tot[1], tot[2], tot[3]);
    *reinterpret_cast<float4*>(o + 4) = make_float4(tot[4], tot[5], tot[6], tot[7]);
  }
}

// ------------------------------------------------------------------------------------------
// ScatteredCacheBuilder step on the device (core/kv_cache.rs:119-237): write slots, start
// positions, advance for active slots, and the RoPE table for the positions AFTER the advance
// (core/batched_transformer.rs:438-447).
// ------------------------------------------------------------------------------------------
__global__ void kv_builder_kernel(uint32_t* __restrict__ pos, uint32_t* __restrict__ idx,
                                  const uint8_t* __restrict__ active, uint32_t* __restrict__ start_pos,
                                  uint32_t* __restrict__ widx, float* __restrict__ rope_cs,
                                  const float* __restrict__ inv_freq, int B, int T, int ctx, int hd,
                                  int rope_before) {
  // rope_before: the non-batched transformer rotates by current_seq_len BEFORE the append (core/transformer.rs:925-926);
  // the batched one reads the builder's positions after indices_and_mask advanced them
  const int b = blockIdx.x;
  if (b >= B) return;
  __shared__ uint32_t s_pos_after;
  if (threadIdx.x == 0) {
    uint32_t p = pos[b], i = idx[b];
    const uint32_t p0 = p;
    start_pos[b] = p;
    for (int t = 0; t < T; ++t) widx[b * T + t] = active[b] ? (uint32_t)((i + t) % ctx) : i;
    if (active[b]) {
      pos[b] = p + T;
      idx[b] = (uint32_t)((i + T) % ctx);
      p += T;
    }
    s_pos_after = rope_before ? p0 : p;
  }
  __syncthreads();
  if (rope_cs) {
    const int half = hd / 2;
    for (int e = threadIdx.x; e < T * half; e += blockDim.x) {
      int t = e / half, i = e % half;
      float ang = (float)(s_pos_after + (uint32_t)t) * inv_freq[i];
      float s, c;
      dsm_sincosf(ang, &s, &c);
      rope_cs[((long)(b * T + t) * half + i) * 2] = c;
      rope_cs[((long)(b * T + t) * half + i) * 2 + 1] = s;
    }
  }
}

// ------------------------------------------------------------------------------------------
// ASR glue — core/asr.rs:165-189 (token shift, pad select) + core/lm.rs:983-995 (embedding sum)
// one workgroup per slot.
// ------------------------------------------------------------------------------------------
__global__ void lm_input_kernel(float* __restrict__ x, const uint16_t* __restrict__ text_emb,
                                const uint16_t* __restrict__ audio_emb /* [nc][audio_vocab][d] */,
                                const uint32_t* __restrict__ codes /* [B][nc] */, uint32_t* __restrict__ next_cb,
                                const uint32_t* __restrict__ text_token, const uint8_t* __restrict__ first_step,
                                const uint8_t* __restrict__ active, int nc, int d, int audio_vocab,
                                uint32_t pad_tok, uint32_t start_tok) {
  // grid (B, d / 512): each thread sums 2 adjacent columns over the 1 + nc tables (independent 4-byte loads)
  const int b = blockIdx.x;
  __shared__ uint32_t toks[64];
  const bool first = first_step[b] != 0;
  if (threadIdx.x < nc) toks[threadIdx.x] = first ? pad_tok : next_cb[b * nc + threadIdx.x];
  __syncthreads();
  const uint32_t tt = first ? start_tok : text_token[b];
  const int j = (blockIdx.y * blockDim.x + threadIdx.x) * 2;
  if (j < d) {
    uint32_t t2 = *reinterpret_cast<const uint32_t*>(text_emb + (long)tt * d + j);
    float e0 = __uint_as_float(t2 << 16), e1 = __uint_as_float(t2 & 0xFFFF0000u);
    for (int i = 0; i < nc; ++i) {  // emb = emb + e, in codebook order — core/lm.rs:988-993
      uint32_t a2 = *reinterpret_cast<const uint32_t*>(audio_emb + ((long)i * audio_vocab + toks[i]) * d + j);
      e0 = e0 + __uint_as_float(a2 << 16);
      e1 = e1 + __uint_as_float(a2 & 0xFFFF0000u);
    }
    *reinterpret_cast<float2*>(x + (long)b * d + j) = make_float2(e0, e1);
  }
}

// next_codebooks = mask ? new codes : old (core/asr.rs:177-183); runs after lm_input_kernel consumed the old values
__global__ void lm_next_codebooks_kernel(const uint32_t* __restrict__ codes, uint32_t* __restrict__ next_cb,
                                         const uint8_t* __restrict__ active, int B, int nc) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B * nc && active[i / nc]) next_cb[i] = codes[i];
}

// first-occurrence argmax of one row by a 256-thread block; valid in every thread after the call
__device__ inline int block_argmax_first(const float* __restrict__ lg, int V) {
  float bv = -DSM_INF_F;
  int bi = 0x7FFFFFFF;
  for (int j = threadIdx.x; j < V; j += blockDim.x) {
    float v = lg[j];
    if (v > bv || (v == bv && j < bi)) { bv = v; bi = j; }
  }
  __shared__ float sv[256];
  __shared__ int si[256];
  sv[threadIdx.x] = bv;
  si[threadIdx.x] = bi;
  __syncthreads();
  for (int off = blockDim.x / 2; off >= 1; off >>= 1) {
    if ((int)threadIdx.x < off) {
      float ov = sv[threadIdx.x + off];
      int oi = si[threadIdx.x + off];
      if (ov > sv[threadIdx.x] || (ov == sv[threadIdx.x] && oi < si[threadIdx.x])) {
        sv[threadIdx.x] = ov;
        si[threadIdx.x] = oi;
      }
    }
    __syncthreads();
  }
  const int r = si[0] == 0x7FFFFFFF ? 0 : si[0];  // all-NaN row: index 0, like a `>` scan from element 0
  __syncthreads();
  return r;
}

// argmax over the text logits (first occurrence on ties, core/asr.rs:208-210) + item-state update
__global__ void lm_argmax_kernel(const float* __restrict__ logits, int V, uint32_t* __restrict__ text_out,
                                 uint32_t* __restrict__ text_token, uint8_t* __restrict__ first_step,
                                 const uint8_t* __restrict__ active) {
  const int b = blockIdx.x;
  const uint32_t tok = (uint32_t)block_argmax_first(logits + (long)b * V, V);
  if (threadIdx.x == 0) {
    text_out[b] = tok;
    if (active[b]) {
      text_token[b] = tok;
      first_step[b] = 0;
    }
  }
}

// temperature > 0 (r04): candle_nn::sampling::gumbel_softmax of the text logits — core/asr.rs:211-215, dsm_sampling.h
// dsm_gumbel_value — with one seeded ChaCha12 stream per slot: entry j of the step takes word pos + j (pos a multiple of 16: a
// thread produces whole blocks), argmax with first occurrence on ties; an active slot's stream advances by V rounded up to 16.
__global__ void lm_gumbel_kernel(const float* __restrict__ logits, int V, float temperature, const uint32_t* __restrict__ rng_key,
                                 unsigned long long* __restrict__ rng_pos, uint32_t* __restrict__ text_out,
                                 uint32_t* __restrict__ text_token, uint8_t* __restrict__ first_step,
                                 const uint8_t* __restrict__ active) {
  const int b = blockIdx.x;
  const float* lg = logits + (long)b * V;
  uint32_t key[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) key[i] = rng_key[b * 8 + i];
  const unsigned long long pos = rng_pos[b];
  float bv = -DSM_INF_F;
  int bi = 0x7FFFFFFF;
  const int nblk = (V + 15) >> 4;
  for (int blk = threadIdx.x; blk < nblk; blk += blockDim.x) {
    uint32_t w[16];
    dsm_chacha_block(key, (pos >> 4) + (unsigned long long)blk, 12, w);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int j = 16 * blk + i;
      if (j < V) {
        const float v = dsm_gumbel_value(lg[j], w[i], temperature);
        if (v > bv || (v == bv && j < bi)) { bv = v; bi = j; }
      }
    }
  }
  __shared__ float sv[256];
  __shared__ int si[256];
  sv[threadIdx.x] = bv;
  si[threadIdx.x] = bi;
  __syncthreads();
  for (int off = blockDim.x / 2; off >= 1; off >>= 1) {
    if ((int)threadIdx.x < off) {
      const float ov = sv[threadIdx.x + off];
      const int oi = si[threadIdx.x + off];
      if (ov > sv[threadIdx.x] || (ov == sv[threadIdx.x] && oi < si[threadIdx.x])) { sv[threadIdx.x] = ov; si[threadIdx.x] = oi; }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const uint32_t tok = si[0] == 0x7FFFFFFF ? 0u : (uint32_t)si[0];
    text_out[b] = tok;
    if (active[b]) {
      text_token[b] = tok;
      first_step[b] = 0;
      rng_pos[b] = pos + 16ull * (unsigned long long)nblk;
    }
  }
}

// ------------------------------------------------------------------------------------------
// TTS glue — core/tts_streaming.rs:117-242, core/lm.rs:640-684, :983-995
// ------------------------------------------------------------------------------------------
// tokens [B][1+nc] int32: text token, then one entry per codebook; -1 = None ("literal zeros": the embedding is skipped)
__global__ void tts_input_kernel(float* __restrict__ x, const uint16_t* __restrict__ text_emb,
                                 const uint16_t* __restrict__ audio_emb /* [nc][audio_vocab][d] */,
                                 const int32_t* __restrict__ tokens, int nc, int d, int audio_vocab) {
  const int b = blockIdx.x;
  __shared__ int32_t toks[72];
  if ((int)threadIdx.x < 1 + nc) toks[threadIdx.x] = tokens[b * (1 + nc) + threadIdx.x];
  __syncthreads();
  const int j = (blockIdx.y * blockDim.x + threadIdx.x) * 2;
  if (j < d) {
    uint32_t t2 = *reinterpret_cast<const uint32_t*>(text_emb + (long)toks[0] * d + j);
    float e0 = __uint_as_float(t2 << 16), e1 = __uint_as_float(t2 & 0xFFFF0000u);
    for (int i = 0; i < nc; ++i) {
      const int32_t t = toks[1 + i];
      if (t < 0) continue;
      uint32_t a2 = *reinterpret_cast<const uint32_t*>(audio_emb + ((long)i * audio_vocab + t) * d + j);
      e0 = e0 + __uint_as_float(a2 << 16);
      e1 = e1 + __uint_as_float(a2 & 0xFFFF0000u);
    }
    *reinterpret_cast<float2*>(x + (long)b * d + j) = make_float2(e0, e1);
  }
}

// Seeded top-k sampling of one row by a 256-thread block (dsm_sampling.h: LogitsProcessor with Sampling::TopK).
// keys: LDS, vpad (power of two >= V) u64 entries; scratch: LDS, 9 floats.  Returns the token in every thread and
// advances the slot's generator by one word.  Reductions follow the canonical block order (thread t owns j = t mod 256
// ascending, wave butterflies, four wave totals left to right); the bitonic sort orders unique keys, so its network
// shape does not matter.
struct SampleArgs {
  const int32_t* top_k;   // [B]: 0 = ArgMax
  const float* inv_t;     // [B]: (float)(1 / temperature)
  const uint32_t* key;    // [B][8] ChaCha key
  uint32_t* pos;          // [B] words drawn so far from this processor's stream
  int vpad;
};
__device__ inline uint32_t block_sample_topk(const float* __restrict__ lg, int V, int k, float invT,
                                             const uint32_t* __restrict__ key, uint32_t* __restrict__ pos, int vpad,
                                             uint64_t* keys, float* scratch) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float m = -DSM_INF_F;
  for (int j = tid; j < V; j += 256) m = fmaxf(m, lg[j] * invT);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  if (lane == 0) scratch[wave] = m;
  __syncthreads();
  m = fmaxf(fmaxf(scratch[0], scratch[1]), fmaxf(scratch[2], scratch[3]));
  float tp = 0.0f;
  for (int j = tid; j < V; j += 256) {
    const float e = dsm_expf(lg[j] * invT - m);
    keys[j] = (uint64_t)__float_as_uint(e);
    tp = tp + e;
  }
  tp = wave_sum64(tp);
  if (lane == 0) scratch[4 + wave] = tp;
  __syncthreads();
  const float sum = ((scratch[4] + scratch[5]) + scratch[6]) + scratch[7];
  for (int j = tid; j < vpad; j += 256)
    keys[j] = j < V ? dsm_sample_key(__uint_as_float((uint32_t)keys[j]) / sum, (uint32_t)j) : 0ull;
  __syncthreads();
  if (k < V) {  // bitonic sort, descending
    for (int size = 2; size <= vpad; size <<= 1)
      for (int stride = size >> 1; stride > 0; stride >>= 1) {
        for (int i = tid; i < (vpad >> 1); i += 256) {
          const int lo = 2 * i - (i & (stride - 1));
          const int hi = lo + stride;
          const bool desc = (lo & size) == 0;
          const uint64_t a = keys[lo], b = keys[hi];
          if ((a < b) == desc) { keys[lo] = b; keys[hi] = a; }
        }
        __syncthreads();
      }
  }
  if (tid == 0) {
    const uint32_t u = dsm_chacha_word(key, (uint64_t)*pos, 12);
    *pos = *pos + 1u;
    scratch[8] = __uint_as_float(dsm_weighted_draw(keys, k < V ? k : V, u));
  }
  __syncthreads();
  const uint32_t tok = __float_as_uint(scratch[8]);
  __syncthreads();
  return tok;
}

// text-token rule of State::step (:178-197).  allowed >= 0: Text(v); -1: Pad; -2: PadOrEpad (text_lp.sample of the text
// logits — argmax, or seeded top-k for slots configured with dsm_tts_set_sampling — decides pad vs end-of-pad unless
// force_eop, i.e. consecutive_pads > max_consecutive_pads)
__global__ void tts_text_token_kernel(const float* __restrict__ logits, int V, const int32_t* __restrict__ allowed,
                                      const uint8_t* __restrict__ force_eop, uint32_t pad, uint32_t eop,
                                      uint32_t* __restrict__ text_token, uint32_t* __restrict__ last_tok, SampleArgs sa,
                                      const uint8_t* __restrict__ active) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int b = blockIdx.x;
  if (!active[b]) return;  // a paused slot takes no step: in particular its text_lp draws nothing (r03: it used to advance the slot's generator)
  const int32_t al = allowed[b];
  uint32_t tok;
  if (al >= 0) tok = (uint32_t)al;
  else if (al == -1) tok = pad;
  else if (force_eop[b]) tok = eop;
  else {  // block-uniform branch
    uint32_t sampled;
    if (sa.top_k && sa.top_k[b] > 0)
      sampled = block_sample_topk(logits + (long)b * V, V, sa.top_k[b], sa.inv_t[b], sa.key + 8 * b, sa.pos + b, sa.vpad,
                                  reinterpret_cast<uint64_t*>(smem), reinterpret_cast<float*>(smem + 8 * (size_t)sa.vpad));
    else
      sampled = (uint32_t)block_argmax_first(logits + (long)b * V, V);
    tok = (sampled == pad) ? pad : eop;
  }
  if (threadIdx.x == 0) {
    text_token[b] = tok;
    last_tok[b] = tok;
  }
}

// e[row] = table[last_tok[row / rps]] (LowRankEmbeddings::forward with the low-rank product folded into the table at load);
// rps batch rows per slot: both rows of a guided slot take the slot's token (core/lm.rs:702-709)
__global__ void dep_gather_kernel(float* __restrict__ e, const float* __restrict__ table,
                                  const uint32_t* __restrict__ last_tok, int vocab, int D, int rps) {
  const int b = blockIdx.x;
  uint32_t t = last_tok[b / rps];
  if (t >= (uint32_t)vocab) t = 0;
  const float4* src = reinterpret_cast<const float4*>(table + (long)t * D);
  float4* dst = reinterpret_cast<float4*>(e + (long)b * D);
  for (int j = threadIdx.x; j < D / 4; j += blockDim.x) dst[j] = src[j];
}

// DepFormer slice head (r03): what dep_gather_kernel + the depformer_in GEMM's reduce/residual epilogue + kv_builder_kernel
// (T = 1, no RoPE) + row_norm_kernel did in five launches, for one batch row per workgroup (256 threads):
//   x  = emb_k[last token] + proj[row][g*D ...]   (core/lm.rs:655-661: depformer_in(xs) + emb; proj = ALL weight groups' input
//        projections of the main LM's output, one GEMM per step — they do not depend on the slice)
//   ring bookkeeping of ScatteredCacheBuilder for one new position (kv_builder_kernel's thread 0)
//   xn = norm1 of layer 0 over x, canonical row reduction (row_norm_kernel's arithmetic)
// `rv + o` of the GEMM epilogue is a single f32 add either way: same bits as the five launches.
struct DepHeadArgs {
  float* x;             // [rows][D] the slice's input (null: no head — dep_argmax_kernel's last slice)
  float* xn;            // [rows][D] norm1 of layer 0 over x
  const float* table;   // emb_k, [vocab][D]
  int vocab, D, rps;
  const float* proj;    // this slice's weight group inside the stacked projections, row stride proj_ld
  long proj_ld;
  uint32_t* pos;
  uint32_t* idx;
  const uint8_t* active;  // per batch row
  uint32_t* start_pos;
  uint32_t* widx;
  int ctx;
  const float* nw;
  const float* nb;
  float eps;
  int rms;
};
// one batch row b with the slot's token t; all 256 threads; `red` = 8 floats of LDS (a barrier separates two calls)
__device__ __forceinline__ void dep_head_row(const DepHeadArgs& h, int b, uint32_t t, float* red) {
  if (threadIdx.x == 0) {  // kv_builder_kernel, T = 1
    const uint32_t p = h.pos[b], i = h.idx[b];
    h.start_pos[b] = p;
    h.widx[b] = i;  // active: (i + 0) % ctx = i (i < ctx); inactive: i
    if (h.active[b]) {
      h.pos[b] = p + 1;
      h.idx[b] = (uint32_t)((i + 1) % h.ctx);
    }
  }
  if (t >= (uint32_t)h.vocab) t = 0;
  const int D = h.D;
  const float* er = h.table + (long)t * D;
  const float* pr = h.proj + (long)b * h.proj_ld;
  float4 v[DSM_ROW_ITS];
  float s = 0.0f, s2 = 0.0f;
#pragma unroll
  for (int it = 0; it < DSM_ROW_ITS; ++it) {
    const int i = it * 1024 + 4 * (int)threadIdx.x;
    v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < D) {
      const float4 ev = *reinterpret_cast<const float4*>(er + i);
      const float4 pv = *reinterpret_cast<const float4*>(pr + i);
      v[it] = make_float4(ev.x + pv.x, ev.y + pv.y, ev.z + pv.z, ev.w + pv.w);
      *reinterpret_cast<float4*>(h.x + (long)b * D + i) = v[it];
      s = s + v[it].x; s2 = DSM_FMAF(v[it].x, v[it].x, s2);
      s = s + v[it].y; s2 = DSM_FMAF(v[it].y, v[it].y, s2);
      s = s + v[it].z; s2 = DSM_FMAF(v[it].z, v[it].z, s2);
      s = s + v[it].w; s2 = DSM_FMAF(v[it].w, v[it].w, s2);
    }
  }
  block_row_sums(s, s2, red);
  row_norm_apply(v, s, s2, D, h.eps, h.rms, h.nw, h.nb, h.xn + (long)b * D);
}
__global__ __launch_bounds__(256) void dep_head_kernel(DepHeadArgs h, const uint32_t* __restrict__ last_tok) {
  __shared__ float red[8];
  const int b = blockIdx.x;
  dep_head_row(h, b, last_tok[b / h.rps], red);
}

// Classifier-free guidance mix of a slot's two batch rows — core/tts_streaming.rs:166-172, core/lm.rs:718-721:
//   ((l0 * a)? - (l1 * (a - 1.))?)?   with Tensor * f64 = affine(mul, 0.): v * (mul as f32) + 0f32 on the CPU backend.
// out [B][V]; rows [B * 2][V]; slots without guidance copy row 0.  grid (ceil(V / 256), B).
__global__ void cfg_mix_kernel(float* __restrict__ out, const float* __restrict__ rows, int V, const uint8_t* __restrict__ on,
                               const float* __restrict__ fa, const float* __restrict__ fb) {
  const int b = blockIdx.y, j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= V) return;
  const float l0 = rows[(long)(2 * b) * V + j];
  float o = l0;
  if (on[b]) {
    const float l1 = rows[(long)(2 * b + 1) * V + j];
    const float x = l0 * fa[b] + 0.0f, y = l1 * fb[b] + 0.0f;
    o = x - y;
  }
  out[(long)b * V + j] = o;
}

// compute_kv(CaSrc::Tokens) of one batch row — core/transformer.rs:299-318: kv [n][2][H][hd] (the in_proj_kv product) ->
// K, V [H][smax][hd] in the cache dtype (rounded like the ring cache's rows).  kc / vc point at the row's block.
template <typename KVT>
__global__ void ca_kv_scatter_kernel(const float* __restrict__ kv, KVT* __restrict__ kc, KVT* __restrict__ vc, int n, int d,
                                     int hd, int smax) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)n * d) return;
  const int j = (int)(i / d), c = (int)(i % d), h = c / hd, ii = c % hd;
  store_kv(kc + ((long)h * smax + j) * hd + ii, kv[(long)j * 2 * d + c]);
  store_kv(vc + ((long)h * smax + j) * hd + ii, kv[(long)j * 2 * d + d + c]);
}

// DepFormer::sample slice epilogue: lp.sample (ArgMax, or seeded top-k per slot), forced pre-delay pad for the NEXT
// slice's input (:671-680)
// Where a slot's logits come from (r03): the output GEMM's split-K slabs, summed here in canonical order (gemm_reduce_kernel's
// arithmetic for a plain EPI_STORE: no bias, scale, residual), or the finished rows; with two batch rows per slot the
// guidance mix of cfg_mix_kernel is applied on the way.  One launch instead of reduce + mix + sampler per depformer slice.
struct LogitSrc {
  const float* ws;    // non-null: slabs, row m of chunk c at ws + c * cstride + m * ld
  long ld, cstride;
  int chunks;
  const float* rows;  // ws null: [batch rows][V]
  float* store;       // ws non-null, optional: the summed rows are also left here [batch rows][V] (debug tap)
  int rps;            // batch rows per slot
  const uint8_t* cfg_on;
  const float* fa;
  const float* fb;
};
__device__ __forceinline__ f32x4 logit4(const LogitSrc& s, int row, int V, int j) {
  if (s.ws) return slab_sum(s.ws + (long)row * s.ld + j, s.cstride, s.chunks);
  return *reinterpret_cast<const f32x4*>(s.rows + (long)row * V + j);
}

// r04: the sampler's workgroup also runs the NEXT slice's head for its slot's batch rows (dep_head_row: embedding of the token it
// just chose + that slice's projection, ring bookkeeping, norm1) — one launch less per slice; slots that are not running keep
// their last token, as the separate head launch found it.
__global__ __launch_bounds__(256) void dep_argmax_kernel(LogitSrc src, int V, int k, int S, uint32_t* __restrict__ lat,
                                                          uint32_t* __restrict__ last_tok, const uint8_t* __restrict__ run,
                                                          const uint8_t* __restrict__ forced, uint32_t pad, SampleArgs sa, int lds_row_off,
                                                          DepHeadArgs next) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ uint32_t s_next;
  __shared__ float hred[8];
  const int b = blockIdx.x;
  if (!run[b]) {
    if (!next.x) return;
    if (threadIdx.x == 0) s_next = last_tok[b];
  } else {
  float* logits = reinterpret_cast<float*>(smem + lds_row_off);  // the slot's row, staged once (V % 4 == 0)
  const bool mix = src.rps == 2 && src.cfg_on[b];
  if (src.ws || (V & 3) == 0) {
    for (int j = 4 * (int)threadIdx.x; j < V; j += 4 * (int)blockDim.x) {
      f32x4 o = logit4(src, b * src.rps, V, j);
      if (src.ws && src.store) *reinterpret_cast<f32x4*>(src.store + (long)(b * src.rps) * V + j) = o;
      if (mix) {  // cfg_mix_kernel: (l0 * a + 0) - (l1 * (a - 1) + 0)
        const f32x4 l1 = logit4(src, b * src.rps + 1, V, j);
        if (src.ws && src.store) *reinterpret_cast<f32x4*>(src.store + (long)(b * src.rps + 1) * V + j) = l1;
        const float fa = src.fa[b], fb = src.fb[b];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float x = o[i] * fa + 0.0f, y = l1[i] * fb + 0.0f;
          o[i] = x - y;
        }
      }
      *reinterpret_cast<f32x4*>(logits + j) = o;
    }
  } else {  // finished rows of a width that is no multiple of four
    for (int j = threadIdx.x; j < V; j += blockDim.x) {
      float o = src.rows[(long)(b * src.rps) * V + j];
      if (mix) {
        const float x = o * src.fa[b] + 0.0f, y = src.rows[(long)(b * src.rps + 1) * V + j] * src.fb[b] + 0.0f;
        o = x - y;
      }
      logits[j] = o;
    }
  }
  __syncthreads();
  uint32_t tok;
  if (sa.top_k && sa.top_k[b] > 0)
    tok = block_sample_topk(logits, V, sa.top_k[b], sa.inv_t[b], sa.key + 8 * b, sa.pos + b, sa.vpad,
                            reinterpret_cast<uint64_t*>(smem), reinterpret_cast<float*>(smem + 8 * (size_t)sa.vpad));
  else
    tok = (uint32_t)block_argmax_first(logits, V);
  if (threadIdx.x == 0) {
    lat[(long)b * S + k] = tok;
    const uint32_t nt = (forced[b] && k > 0) ? pad : tok;
    last_tok[b] = nt;
    s_next = nt;
  }
  if (!next.x) return;
  }
  __syncthreads();
  const uint32_t t = s_next;
  for (int j = 0; j < next.rps; ++j) {
    if (j > 0) __syncthreads();  // hred is reused
    dep_head_row(next, b * next.rps + j, t, hred);
  }
}

// extra heads: f32 softmax over `dim` classes, class-0 probability -> prs[head][slot] (core/asr.rs:195-203)
__global__ void extra_heads_kernel(const float* __restrict__ eh /* [B][nh*dim] */, float* __restrict__ prs, int B,
                                   int nh, int dim, int prs_stride /* slots of the whole batch */) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * nh) return;
  int b = i / nh, h = i % nh;
  const float* lg = eh + ((long)b * nh + h) * dim;
  float m = lg[0];
  for (int k = 1; k < dim; ++k) m = lg[k] > m ? lg[k] : m;
  float sum = 0.0f, e0 = 0.0f;
  for (int k = 0; k < dim; ++k) {
    float e = dsm_expf(lg[k] - m);
    if (k == 0) e0 = e;
    sum = sum + e;
  }
  prs[(long)h * prs_stride + b] = e0 / sum;
}

// ------------------------------------------------------------------------------------------
// SEANet front end, fused (r02): the encoder's first three convs for one slot and 64 output frames in one workgroup —
//   init conv (1 -> 64, k 7)  ->  [ELU -> conv k 3 (64 -> 32) -> ELU -> conv k 1 (32 -> 64)] + skip  ->  ELU
// (core/seanet.rs:292-302 with SeaNetResnetBlock::step :140-150; core/conv.rs:335-367 for the carried frames).  As three
// GEMM launches these layers move ~7 GB at the capacity batch for 65 GFLOP of work (K = 7 / 192 / 32, N = 64 / 32 / 64
// over B x 1920 rows: 3.3 ms of a 69 ms step at B = 2048); here the 64-channel intermediates never leave LDS: the
// workgroup reads 72 PCM samples and writes one [64][64] tile of the next conv's input.
// Arithmetic is the GEMM path's, operation for operation (dsm_numerics.h):
//   * init conv: K = 7 sits in lane group q = 0 of one 32-block, so the canonical chain is fmaf over taps 0..6 from +0
//     (the zero-padded taps add +0 to an accumulator that is never -0), then + bias: VALU, no MFMA;
//   * conv k 3: K = 192 = six 32-blocks of one chunk, v_mfma_f32_16x16x4_f32 in the canonical order
//     (k = 32 blk + 8 q + s, s outer), + bias, ELU; its im2col rows are windows of the ELU'd tile in LDS;
//   * conv k 1: one 32-block, + bias, skip + value (the raw init-conv output, kept in LDS), ELU.
// Carried frames: the k-3 conv's two past frames are read from its concat buffer's prefix (written by
// conv_state_shift_kernel after the previous step) by the first tile of a slot; the last tile leaves the slot's two newest
// ELU'd frames in that buffer's tail for the shift to carry.  The buffers in between (raw init output, the k-1 conv's
// input) are not written at all.
// ------------------------------------------------------------------------------------------
struct SeanetFrontArgs {
  const float* cat_init;  // [B][S0 + T] PCM with S0 = 6 carried samples in front
  const float *w0, *b0;   // [64][ld0] (7 taps used), bias or null
  const float *w1, *b1;   // [32][ld1], k = tap * 64 + channel
  const float *w2, *b2;   // [64][ld2]
  float* cat_ra;          // [B][2 + T][64]: ELU'd init output as the k-3 conv sees it (prefix read, tail written)
  float* cat_down;        // [B][Sd + T][64]: the strided conv's concat buffer, rows Sd.. written
  int ld0, ld1, ld2;
  int T, S0, Sd;
};

template <int TM>
constexpr size_t seanet_front_lds() { return sizeof(float) * ((TM + 2) * 64 + TM * 64 + TM * 32 + (TM + 2 + 7 - 1)); }
template <int TM>  // frames per workgroup: 64 or 32 (smaller tiles: more workgroups per CU to hide the three phases behind each other)
__global__ __launch_bounds__(256) void seanet_front_kernel(SeanetFrontArgs a) {
  constexpr int C0 = 64, C1 = 32, TAPS0 = 7, MTILES = TM / 16;
  static_assert(TM == 64 || TM == 32, "tile");
  // Dynamic LDS (seanet_front_lds<TM>() bytes), not static arrays: a static 41 KB tells the compiler that three workgroups fit on a
  // CU and it pads the register allocation up to that occupancy (136 VGPRs for 88 used) — registers the LM streams' workgroups
  // can use beside this kernel (r04, see gemm_bx3u_kernel).
  extern __shared__ __attribute__((aligned(16))) float front_lds[];
  float* e0 = front_lds;                    // [(TM + 2) * C0] ELU(init conv), frames t0-2 .. t0+TM-1; 16-byte unit u of row R at u ^ (R & 15)
  float* y0 = e0 + (TM + 2) * C0;           // [TM * C0] raw init conv (the skip), same swizzle
  float* hs = y0 + TM * C0;                 // [TM * C1] ELU(conv k 3 + bias): the k-1 conv's activation block (dsm_xs_sw swizzle)
  float* xs = hs + TM * C1;                 // [TM + 2 + TAPS0 - 1]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int b = blockIdx.y, t0 = blockIdx.x * TM;
  const bool first = t0 == 0, last = t0 + TM == a.T;
  const float* cat = a.cat_init + (long)b * (a.S0 + a.T);
  // e0 row R (frame t0 - 2 + R) needs cat[t0 - 2 + R + s], s = 0..6
  for (int i = tid; i < TM + 2 + TAPS0 - 1; i += 256) {
    const int c = t0 - 2 + i;
    xs[i] = c >= 0 ? cat[c] : 0.0f;  // c < 0 only feeds rows 0 and 1 of the first tile, which come from the carried frames
  }
  // ---- init conv + ELU (VALU): a thread keeps its four channels' taps and biases in registers and walks rows tid / 16 + 16 i ----
  const int cq = tid & 15;
  float w0r[4][TAPS0], b0r[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
#pragma unroll
    for (int sidx = 0; sidx < TAPS0; ++sidx) w0r[j][sidx] = a.w0[(long)(4 * cq + j) * a.ld0 + sidx];
    b0r[j] = a.b0 ? a.b0[4 * cq + j] : 0.0f;
  }
  __syncthreads();
  for (int R = tid >> 4; R < TM + 2; R += 16) {
    float4 ev;
    if (first && R < 2) {
      ev = *reinterpret_cast<const float4*>(a.cat_ra + ((long)b * (2 + a.T) + R) * C0 + 4 * cq);
    } else {
      float xr[TAPS0];
#pragma unroll
      for (int sidx = 0; sidx < TAPS0; ++sidx) xr[sidx] = xs[R + sidx];
      float yv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float acc = 0.0f;
#pragma unroll
        for (int sidx = 0; sidx < TAPS0; ++sidx) acc = DSM_FMAF(w0r[j][sidx], xr[sidx], acc);
        yv[j] = a.b0 ? acc + b0r[j] : acc;
      }
      ev = make_float4(dsm_elu(yv[0]), dsm_elu(yv[1]), dsm_elu(yv[2]), dsm_elu(yv[3]));
      if (R >= 2) *reinterpret_cast<float4*>(&y0[(R - 2) * C0 + 4 * (cq ^ ((R - 2) & 15))]) = make_float4(yv[0], yv[1], yv[2], yv[3]);
      if (last && R >= TM)  // frames T-2, T-1: what conv_state_shift_kernel carries into the next step
        *reinterpret_cast<float4*>(a.cat_ra + ((long)b * (2 + a.T) + 2 + (t0 - 2 + R)) * C0 + 4 * cq) = ev;
    }
    *reinterpret_cast<float4*>(&e0[R * C0 + 4 * (cq ^ (R & 15))]) = ev;
  }
  __syncthreads();
  // ---- conv k 3: [TM frames][K = 192] x [32][192]^T.  TM = 64: wave -> (n-tile wave & 1, m-tiles 2 (wave >> 1) + {0, 1});
  //      TM = 32: wave -> (n-tile wave & 1, m-tile wave >> 1) ----
  {
    constexpr int MPW = MTILES / 2;  // m-tiles per wave
    const int nt = wave & 1, mh = wave >> 1;
    f32x4 acc[MPW];
#pragma unroll
    for (int m2 = 0; m2 < MPW; ++m2) acc[m2] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* wrow = a.w1 + (long)(16 * nt + r) * a.ld1 + 8 * q;
#pragma unroll
    for (int blk = 0; blk < 6; ++blk) {
      float wa[8], xb[MPW][8];
      load_w8<float>(wrow + 32 * blk, wa);
      const int tap = blk >> 1, u0 = (blk & 1) * 8 + 2 * q;  // k = 32 blk + 8 q: tap k / 64, channel k % 64
#pragma unroll
      for (int m2 = 0; m2 < MPW; ++m2) {
        const int R = 16 * (MPW * mh + m2) + r + tap;
        const float4 f0 = *reinterpret_cast<const float4*>(&e0[R * C0 + 4 * (u0 ^ (R & 15))]);
        const float4 f1 = *reinterpret_cast<const float4*>(&e0[R * C0 + 4 * ((u0 + 1) ^ (R & 15))]);
        xb[m2][0] = f0.x; xb[m2][1] = f0.y; xb[m2][2] = f0.z; xb[m2][3] = f0.w;
        xb[m2][4] = f1.x; xb[m2][5] = f1.y; xb[m2][6] = f1.z; xb[m2][7] = f1.w;
      }
#pragma unroll
      for (int sidx = 0; sidx < 8; ++sidx)
#pragma unroll
        for (int m2 = 0; m2 < MPW; ++m2) acc[m2] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[sidx], xb[m2][sidx], acc[m2], 0, 0, 0);
    }
#pragma unroll
    for (int m2 = 0; m2 < MPW; ++m2) {
      const int m = 16 * (MPW * mh + m2) + r, n = 16 * nt + 4 * q;  // the lane holds channels n..n+3 of frame m
      float o[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = dsm_elu(a.b1 ? acc[m2][i] + a.b1[n + i] : acc[m2][i]);
      *reinterpret_cast<float4*>(&hs[m * C1 + 4 * ((n >> 2) ^ dsm_xs_sw(m))]) = make_float4(o[0], o[1], o[2], o[3]);
    }
  }
  __syncthreads();
  // ---- conv k 1 + skip + ELU: [TM frames][32] x [64][32]^T; wave -> n-tile `wave`, every m-tile ----
  {
    f32x4 acc[MTILES];
#pragma unroll
    for (int mt = 0; mt < MTILES; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float wa[8], xb[MTILES][8];
    load_w8<float>(a.w2 + (long)(16 * wave + r) * a.ld2 + 8 * q, wa);
    const int xu0 = 4 * ((2 * q) ^ dsm_xs_sw(r)), xu1 = xu0 ^ 4;
#pragma unroll
    for (int mt = 0; mt < MTILES; ++mt) {
      const float* fp = &hs[(16 * mt + r) * C1];
      const float4 f0 = *reinterpret_cast<const float4*>(fp + xu0), f1 = *reinterpret_cast<const float4*>(fp + xu1);
      xb[mt][0] = f0.x; xb[mt][1] = f0.y; xb[mt][2] = f0.z; xb[mt][3] = f0.w;
      xb[mt][4] = f1.x; xb[mt][5] = f1.y; xb[mt][6] = f1.z; xb[mt][7] = f1.w;
    }
#pragma unroll
    for (int sidx = 0; sidx < 8; ++sidx)
#pragma unroll
      for (int mt = 0; mt < MTILES; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[sidx], xb[mt][sidx], acc[mt], 0, 0, 0);
    float* out = a.cat_down + ((long)b * (a.Sd + a.T) + a.Sd + t0) * C0;
#pragma unroll
    for (int mt = 0; mt < MTILES; ++mt) {
      const int m = 16 * mt + r, n = 16 * wave + 4 * q;
      const float4 rv = *reinterpret_cast<const float4*>(&y0[m * C0 + 4 * ((n >> 2) ^ (m & 15))]);
      float o[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = a.b2 ? acc[mt][i] + a.b2[n + i] : acc[mt][i];
      o[0] = rv.x + o[0]; o[1] = rv.y + o[1]; o[2] = rv.z + o[2]; o[3] = rv.w + o[3];
      *reinterpret_cast<float4*>(out + (long)m * C0 + n) = make_float4(dsm_elu(o[0]), dsm_elu(o[1]), dsm_elu(o[2]), dsm_elu(o[3]));
    }
  }
}

// ------------------------------------------------------------------------------------------
// Streaming-conv state: in-place shift of the consumer's concat buffer after it ran
// (core/conv.rs:335-367).  desc[i] = {ptr, state_len S, step frames T, channels C, batch stride}.
// active slots: cat[b][0..S) = cat[b][T..T+S); inactive: unchanged (zeroed on the first call).
// ------------------------------------------------------------------------------------------
struct ConvStateDesc {
  float* cat;
  long bstride;
  int S, T, C, replicate;
};

__global__ void conv_state_shift_kernel(const ConvStateDesc* __restrict__ descs, const uint8_t* __restrict__ active,
                                        int first_call) {
  const ConvStateDesc dsc = descs[blockIdx.y];
  const int b = blockIdx.x;
  float* base = dsc.cat + (long)b * dsc.bstride;
  const int n = dsc.S * dsc.C;
  if (active[b]) {
    const float* src = base + (long)dsc.T * dsc.C;
    // S <= T for every conv of the model, so source and destination do not overlap
    for (int i = threadIdx.x; i < n; i += blockDim.x) base[i] = src[i];
  } else if (first_call) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) base[i] = 0.0f;
  }
}

// first call of a replicate-padded conv (ConvDownsample1d, core/conv.rs:318-327,530): the left pad
// repeats the first input frame of every slot
__global__ void conv_replicate_init_kernel(ConvStateDesc dsc) {
  const int b = blockIdx.x;
  float* base = dsc.cat + (long)b * dsc.bstride;
  const float* first = base + (long)dsc.S * dsc.C;
  for (int i = threadIdx.x; i < dsc.S * dsc.C; i += blockDim.x) base[i] = first[i % dsc.C];
}

__global__ void conv_state_reset_kernel(const ConvStateDesc* __restrict__ descs, int slot) {
  const ConvStateDesc dsc = descs[blockIdx.x];
  float* base = dsc.cat + (long)slot * dsc.bstride;
  for (int i = threadIdx.x; i < dsc.S * dsc.C; i += blockDim.x) base[i] = 0.0f;
}

// ------------------------------------------------------------------------------------------
// RVQ select: reduce the per-tile (dist, idx) partials, emit the code, update the residual
// (core/quantization.rs:219-229).  One workgroup per row.
// ------------------------------------------------------------------------------------------
__global__ void rvq_select_kernel(const float* __restrict__ pval, const uint32_t* __restrict__ pidx, int n_tiles,
                                  int M, uint32_t* __restrict__ codes, int code_stride, int code_off,
                                  float* __restrict__ residual, const float* __restrict__ E, int dim, int ldE) {
  const int m = blockIdx.x;
  __shared__ uint32_t s_code;
  if (threadIdx.x < 64) {
    float bv = DSM_INF_F;
    uint32_t bi = 0xFFFFFFFFu;
    for (int t = threadIdx.x; t < n_tiles; t += 64) {
      float v = pval[(long)t * M + m];
      uint32_t i = pidx[(long)t * M + m];
      if (v < bv || (v == bv && i < bi)) { bv = v; bi = i; }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      float ov = __shfl_xor(bv, off, 64);
      uint32_t oi = __shfl_xor(bi, off, 64);
      if (ov < bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (threadIdx.x == 0) {
      s_code = bi;
      codes[(long)m * code_stride + code_off] = bi;
    }
  }
  __syncthreads();
  const uint32_t c = s_code;
  for (int i = threadIdx.x; i < dim; i += blockDim.x)
    residual[(long)m * dim + i] = residual[(long)m * dim + i] - E[(long)c * ldE + i];
}

// ------------------------------------------------------------------------------------------
// Decode side (Mimi::decode_step, core/mimi.rs:217-225)
// ------------------------------------------------------------------------------------------
// ResidualVectorQuantization::decode — core/quantization.rs:231-248: gather E_i[code_i] and sum the layers in order.
// One workgroup per slot; q_first = layer 0 of rvq_first, q_rest = sum over rvq_rest layers.
struct RvqGatherArgs {
  const float* const* emb;  // n_q device pointers: [bins][ldE] f32 (index 0 = rvq_first, 1.. = rvq_rest)
  int n_q, bins, dim, ldE;
};
__global__ void rvq_gather_sum_kernel(RvqGatherArgs a, const uint32_t* __restrict__ codes, float* __restrict__ q_first,
                                      float* __restrict__ q_rest) {
  const int b = blockIdx.x;
  for (int dd = threadIdx.x; dd < a.dim; dd += blockDim.x) {
    uint32_t c0 = codes[b * a.n_q];
    c0 = c0 < (uint32_t)a.bins ? c0 : 0u;
    q_first[(long)b * a.dim + dd] = a.emb[0][(long)c0 * a.ldE + dd];
    float acc = 0.0f;
    for (int i = 1; i < a.n_q; ++i) {
      uint32_t c = codes[b * a.n_q + i];
      c = c < (uint32_t)a.bins ? c : 0u;
      float ev = a.emb[i][(long)c * a.ldE + dd];
      acc = (i == 1) ? ev : acc + ev;
    }
    if (a.n_q > 1) q_rest[(long)b * a.dim + dd] = acc;
  }
}

// ConvTrUpsample1d::step — depthwise ConvTranspose1d k = 2*stride, one input frame per step, no bias
// (core/conv.rs:558-606, :448-501).  x [B][C]; w [k][C]; carry [B][k - s][C]; y [B][s][C].
__global__ void upsample_dw_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ carry,
                                   float* __restrict__ y, const uint8_t* __restrict__ active, int C, int s, int k,
                                   int has_state) {
  const int b = blockIdx.x;
  for (int i = threadIdx.x; i < s * C; i += blockDim.x) {
    const int o = i / C, c = i % C;
    const float xv = x[(long)b * C + c];
    float v = xv * w[(long)o * C + c];
    if (has_state) v = v + carry[((long)b * (k - s) + o) * C + c];
    y[((long)b * s + o) * C + c] = v;
    if (o < k - s) {  // k - s == s: the thread that consumed carry[o] also refreshes it (no cross-thread hazard)
      const float nv = xv * w[(long)(s + o) * C + c];
      float* cp = &carry[((long)b * (k - s) + o) * C + c];
      if (active[b]) *cp = nv;
      else if (!has_state) *cp = 0.0f;
    }
  }
}

// StreamableConvTranspose1d::step epilogue — core/conv.rs:448-501 with k == 2*stride.  The GEMM produced
// z[(b,t)][kk*OC + co] = sum_ci x[b][t][ci] * w[ci][co][kk]; output frame o = t*s + kk (kk < s) is
//   t == 0 : (z[t][kk] + bias) + (carry[kk] - bias)          (first call of the module: no carry term)
//   t  > 0 : (z[t-1][kk+s] + z[t][kk]) + bias
// new carry[j] = z[T-1][j+s] + bias (kept for inactive slots).  Writes y (raw) and/or y2 (ELU copy).
struct OverlapAddArgs {
  const float* z;
  const float* bias;
  float* carry;  // [B][s][OC]
  const uint8_t* active;
  int T, s, OC, has_state;
  float* Y;
  RowMap ymap;  // rows are (b, o), rpb = T*s
  float* Y2;
  RowMap y2map;
};
__global__ void convtr_overlap_add_kernel(OverlapAddArgs a) {
  const int b = blockIdx.y;
  const int per_b = a.T * a.s * (a.OC >> 2);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < per_b; i += gridDim.x * blockDim.x) {
    const int c4 = i % (a.OC >> 2), o = i / (a.OC >> 2);
    const int co = c4 * 4, t = o / a.s, kk = o % a.s;
    const long ldz = (long)2 * a.s * a.OC;
    const float4 zb = *reinterpret_cast<const float4*>(a.z + ((long)b * a.T + t) * ldz + (long)kk * a.OC + co);
    const float4 bs = *reinterpret_cast<const float4*>(a.bias + co);
    float4 v;
    if (t > 0) {
      const float4 za = *reinterpret_cast<const float4*>(a.z + ((long)b * a.T + t - 1) * ldz + (long)(kk + a.s) * a.OC + co);
      v.x = (za.x + zb.x) + bs.x; v.y = (za.y + zb.y) + bs.y; v.z = (za.z + zb.z) + bs.z; v.w = (za.w + zb.w) + bs.w;
    } else {
      v.x = zb.x + bs.x; v.y = zb.y + bs.y; v.z = zb.z + bs.z; v.w = zb.w + bs.w;
      float* cp = a.carry + ((long)b * a.s + kk) * a.OC + co;
      if (a.has_state) {
        const float4 cv = *reinterpret_cast<const float4*>(cp);
        v.x = v.x + (cv.x - bs.x); v.y = v.y + (cv.y - bs.y); v.z = v.z + (cv.z - bs.z); v.w = v.w + (cv.w - bs.w);
      }
      const float4 zl = *reinterpret_cast<const float4*>(a.z + ((long)b * a.T + a.T - 1) * ldz + (long)(kk + a.s) * a.OC + co);
      if (a.active[b])
        *reinterpret_cast<float4*>(cp) = make_float4(zl.x + bs.x, zl.y + bs.y, zl.z + bs.z, zl.w + bs.w);
      else if (!a.has_state)
        *reinterpret_cast<float4*>(cp) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int m = b * a.T * a.s + o;
    if (a.Y) *reinterpret_cast<float4*>(a.Y + a.ymap.off(m) + co) = v;
    if (a.Y2) *reinterpret_cast<float4*>(a.Y2 + a.y2map.off(m) + co) = make_float4(dsm_elu(v.x), dsm_elu(v.y), dsm_elu(v.z), dsm_elu(v.w));
  }
}

__global__ void fill_u32_kernel(uint32_t* p, uint32_t v, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
