// gemm_pers_kernel.h — EXPERIMENT, not part of the product library (see DESIGN.md §6 "what was tried").
// A persistent, statically balanced variant of gemm_tile_kernel.  Measured on MI355X (stt-1b LM shapes, M = 64) it
// was no faster than the one-tile-per-workgroup kernel (QKV 22.3 vs 21.2 us, gate 37.2 vs 37.6 us) while its 66 KB
// of LDS per workgroup stops kernels of other streams from co-residing, so the engine does not use it.
// experiments/gemm_pers_ablate.hip prices its parts (-DDSM_GEMM_ABL=<mask>).
#pragma once
#include "dsm_kernels.h"

// ---- persistent tiled GEMM (K a multiple of the 256-wide chunk, 16-byte aligned X rows) ----
// The work is cut into atoms (K-chunk, group of MPER m-tiles, 64-row n-tile, one 16-row m-tile) of 64*NT MFMAs per
// wave, flattened chunk-major, and every workgroup takes one contiguous, equally long run of atoms: the 2 x 256
// resident workgroups finish together whatever the shape (the one-tile-per-workgroup kernel left 25 % of the chip
// idle whenever tiles / 512 was x.5).  Inside a run, consecutive atoms of the same n-tile form a unit of up to 4
// m-tiles that shares the weight fragments; the whole [64][256] activation chunk sits in LDS and is restaged only
// when the run crosses into another chunk / m-group, so there is no barrier inside the K loop; weight fragments are
// prefetched a whole unit (eight 32-wide blocks) ahead, across unit boundaries.  Same slab layout and epilogues as gemm_tile_kernel.
#define DSM_XC_LD 260
#ifndef DSM_GEMM_ABL  // experiments/gemm_pers_ablate.hip prices the kernel's parts by compiling them out (timing only)
#define DSM_GEMM_ABL 0
#endif
struct PersGeom {
  int chunks, mgroups, ntiles64, mper, atoms;
};

// weight-prefetch depth in 32-wide blocks: a block is 4 (bf16) or 8 (f32) registers per n-tile
template <typename WT, int NT>
struct PersDepth {
  static constexpr int D = (sizeof(WT) == 4 || NT == 2) ? 4 : 8;
};

template <typename WT, typename KVT, int NT, int EPI, int CNT>
__device__ __forceinline__ void pers_unit(const GemmArgs& a, const float* __restrict__ xs,
                                          Raw8<WT> (&rw)[PersDepth<WT, NT>::D][NT], const WT* (&wcur)[NT],
                                          const WT* (&wnext)[NT], int chunk, int chunks, int m_unit, int n_base, int r,
                                          int q) {
  // On entry rw[i] holds this unit's weight block i, i < D (fetched while the previous unit ran); step i consumes
  // rw[i % D] and refills it with block i + D — of this unit, or of the NEXT unit past block 7 — so D blocks
  // (8192 MFMA cycles' worth) of weight bytes are always in flight per wave.  Two blocks ahead kept only ~4 MB in
  // flight chip-wide and the weight stream, not the MFMAs, set the pace.
  // Activation fragments are read from LDS one step ahead of the MFMAs that use them.
  constexpr int D = PersDepth<WT, NT>::D;
  f32x4 acc[NT][CNT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int mt = 0; mt < CNT; ++mt) acc[nt][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float4 xc[CNT][2], xn[CNT][2];
#pragma unroll
  for (int mt = 0; mt < CNT; ++mt) {
    const float* fp = xs + (16 * mt + r) * DSM_XC_LD + 8 * q;
    xc[mt][0] = *reinterpret_cast<const float4*>(fp);
    xc[mt][1] = *reinterpret_cast<const float4*>(fp + 4);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    float wa[NT][8];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) rw[i % D][nt].unpack(wa[nt]);
    if (!(DSM_GEMM_ABL & 2)) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)  // no next unit: wnext == this unit's rows (loaded, never used)
        rw[i % D][nt].load(i + D < 8 ? wcur[nt] + 32 * (i + D) : wnext[nt] + 32 * (i + D - 8));
    }
    if (i < 7) {
#pragma unroll
      for (int mt = 0; mt < CNT; ++mt) {
        const float* fp = xs + (16 * mt + r) * DSM_XC_LD + 32 * (i + 1) + 8 * q;
        xn[mt][0] = *reinterpret_cast<const float4*>(fp);
        xn[mt][1] = *reinterpret_cast<const float4*>(fp + 4);
      }
    }
    __builtin_amdgcn_sched_barrier(0);  // keep the refill and the fragment reads ahead of this step's MFMAs
    if (!(DSM_GEMM_ABL & 8)) {
#pragma unroll
      for (int s = 0; s < 8; ++s) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int mt = 0; mt < CNT; ++mt) {
            const float4 xv = xc[mt][s >> 2];
            const float xe = (s & 3) == 0 ? xv.x : (s & 3) == 1 ? xv.y : (s & 3) == 2 ? xv.z : xv.w;
            acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[nt][s], xe, acc[nt][mt], 0, 0, 0);
          }
        // round-robin over the NT*CNT accumulators: left alone the scheduler chains one accumulator's MFMAs
        // back to back (40-cycle dependent latency + s_nop padding instead of the 32-cycle issue rate)
        if (NT * CNT > 1) __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < CNT; ++mt) acc[nt][mt][0] += wa[nt][mt] * xc[mt][0].x;
    }
    if (i < 7) {
#pragma unroll
      for (int mt = 0; mt < CNT; ++mt) {
        xc[mt][0] = xn[mt][0];
        xc[mt][1] = xn[mt][1];
      }
    }
  }
  if (chunks > 1) {
    const long ld = (long)a.ws_ntiles * 16;
    const long mpad = (long)((a.M + 15) >> 4) * 16;
    if ((DSM_GEMM_ABL & 4) && acc[0][0][0] != 1234.5f) return;
#pragma unroll
    for (int mt = 0; mt < CNT; ++mt) {
      const int m = m_unit + 16 * mt + r;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int n = n_base + nt * a.nt_stride + 4 * q;
        *reinterpret_cast<f32x4*>(a.ws + ((long)chunk * mpad + m) * ld + n) = acc[nt][mt];
      }
    }
    return;
  }
#pragma unroll
  for (int mt = 0; mt < CNT; ++mt) {
    const int m = m_unit + 16 * mt + r;
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
}

template <typename WT, typename KVT, int NT, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_pers_kernel(GemmArgs a, PersGeom gm) {
  __shared__ __attribute__((aligned(16))) float Xs[64][DSM_XC_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const WT* W = reinterpret_cast<const WT*>(a.W);
  const int mtiles = (a.M + 15) >> 4;
  int p = (int)((long)blockIdx.x * gm.atoms / gridDim.x);
  const int pend = (int)((long)(blockIdx.x + 1) * gm.atoms / gridDim.x);

  // atom p -> (chunk, mgroup, ntile64, mi)
#define DSM_DECODE(P, CH, MG, NTI, MI)        \
  {                                           \
    int u_ = (P);                             \
    MI = u_ % gm.mper; u_ /= gm.mper;         \
    NTI = u_ % gm.ntiles64; u_ /= gm.ntiles64; \
    MG = u_ % gm.mgroups;                     \
    CH = u_ / gm.mgroups;                     \
  }
  // unit starting at atom P: CNT = valid m-tiles it spans (0: the whole unit lies beyond M), LEN = atoms consumed
#define DSM_UNIT(P, CH, MG, NTI, MI, CNT, LEN)                       \
  {                                                                  \
    DSM_DECODE(P, CH, MG, NTI, MI)                                   \
    LEN = min(gm.mper - MI, pend - (P));                             \
    const int first_ = MG * gm.mper + MI;                            \
    CNT = max(0, min(LEN, mtiles - first_));                         \
  }
  int chunk, mg, nti, mi, cnt, len;
  // skip leading empty units
  while (p < pend) {
    DSM_UNIT(p, chunk, mg, nti, mi, cnt, len)
    if (cnt > 0) break;
    p += len;
  }
  if (p >= pend) return;
  constexpr int D = PersDepth<WT, NT>::D;
  Raw8<WT> rw[D][NT];
  const WT* wcur[NT];
  const WT* wnext[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const WT* w0 = W + (long)(nti * 64 + 16 * wave + nt * a.nt_stride + r) * a.Kpad + 8 * q + chunk * DSM_KC;
    wcur[nt] = w0;
    wnext[nt] = w0;
#pragma unroll
    for (int i = 0; i < D; ++i) rw[i][nt].load(w0 + 32 * i);
  }
  int cur_chunk = -1, cur_mg = -1;
  while (p < pend) {
    // find the next non-empty unit (for the weight prefetch that runs across the unit boundary)
    int pn = p + len, nchunk = 0, nmg = 0, nnti = 0, nmi = 0, ncnt = 0, nlen = 0;
    while (pn < pend) {
      DSM_UNIT(pn, nchunk, nmg, nnti, nmi, ncnt, nlen)
      if (ncnt > 0) break;
      pn += nlen;
    }
    const bool has_next = pn < pend;
    if (has_next) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        wnext[nt] = W + (long)(nnti * 64 + 16 * wave + nt * a.nt_stride + r) * a.Kpad + 8 * q + nchunk * DSM_KC;
    }
    if (chunk != cur_chunk || mg != cur_mg) {  // (re)stage the activation chunk: rows of m-group mg, columns of chunk
      __syncthreads();
      const int k0 = chunk * DSM_KC;
#pragma unroll
      for (int half = 0; half < 2; ++half) {  // two batches of 8 loads: 32 staging registers instead of 64
        float4 v[8];
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const int piece = (half * 8 + it) * 256 + tid, row = piece >> 6, c4 = piece & 63;
          int m = mg * (16 * gm.mper) + row;
          m = m < a.M ? m : a.M - 1;
          v[it] = (row < 16 * gm.mper && !(DSM_GEMM_ABL & 1)) ? *reinterpret_cast<const float4*>(a.X + a.xmap.off(m) + k0 + 4 * c4)
                                                              : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const int piece = (half * 8 + it) * 256 + tid, row = piece >> 6, c4 = piece & 63;
          *reinterpret_cast<float4*>(&Xs[row][4 * c4]) = v[it];
        }
      }
      __syncthreads();
      cur_chunk = chunk;
      cur_mg = mg;
    }
    const float* xs = &Xs[16 * mi][0];
    const int m_unit = (mg * gm.mper + mi) * 16;
    const int n_base = nti * 64 + 16 * wave;
    switch (cnt) {
      case 4: pers_unit<WT, KVT, NT, EPI, 4>(a, xs, rw, wcur, wnext, chunk, gm.chunks, m_unit, n_base, r, q); break;
      case 3: pers_unit<WT, KVT, NT, EPI, 3>(a, xs, rw, wcur, wnext, chunk, gm.chunks, m_unit, n_base, r, q); break;
      case 2: pers_unit<WT, KVT, NT, EPI, 2>(a, xs, rw, wcur, wnext, chunk, gm.chunks, m_unit, n_base, r, q); break;
      default: pers_unit<WT, KVT, NT, EPI, 1>(a, xs, rw, wcur, wnext, chunk, gm.chunks, m_unit, n_base, r, q); break;
    }
    p = pn;
    chunk = nchunk; mg = nmg; nti = nnti; mi = nmi; cnt = ncnt; len = nlen;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) wcur[nt] = wnext[nt];
  }
#undef DSM_DECODE
#undef DSM_UNIT
}

