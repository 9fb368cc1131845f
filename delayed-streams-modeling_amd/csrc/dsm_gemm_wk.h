// dsm_gemm_wk.h — small-batch GEMM with the WHOLE reduction inside one workgroup (r04).  Included by dsm_kernels.h.
//
// gemm_tile_kernel / gemm_bx3_kernel cut K across workgroups at M <= 64: every 256-wide chunk is its own workgroup, the chunk
// sums travel through HBM as split-K slabs (chunks x M x N floats written, read back once) and a second launch adds them in
// the canonical order and runs the epilogue.  At the batch the headline is quoted on that is three reduce launches per layer and
// stream group, 1.5-1.75 x the weights' bytes per GEMM (profiles/r03/pmc_hbm_traffic.json) and a seam in front of every consumer.
//
// Here the K-parallelism lives INSIDE the workgroup: a workgroup owns 16 NT output columns x 16 MT rows for the whole K, its
// waves own consecutive 256-wide chunks (CPW each), every wave issues all of its weight loads up front (8 blocks x 16 bytes per
// lane and chunk: the in-order vmcnt counter never makes a later load wait behind a deeper window), takes the activations
// straight from L2 into the B operand (lane (r, q) of the matrix instruction holds elements 8q .. 8q+7 of row r of the block:
// 32 contiguous bytes of an f32 row, no LDS staging, no barrier inside the loop) and, in dot_mode 1, splits them into the three
// bf16 pieces in registers.  The chunk sums meet in LDS and are added left to right from +0 by the wave that owns the output
// tile — the same canonical order, the same bits as the slab reduce (dsm_numerics.h, DOT ORDER) — and the epilogue (bias, scale,
// residual, ELU copy / QKV split + RoPE + ring scatter / SiLU gate) runs right there.  No slabs, no reduce launch.
//
// What it costs: every workgroup reads all of its rows' activations (16 MT x K x 4 bytes from L2) where a split-K workgroup
// reads one chunk of them, so the tile is chosen by the launcher to keep that re-read near the weights' own bytes.
#pragma once

typedef unsigned int dsm_u32x4v __attribute__((ext_vector_type(4)));

// eight f32 -> the three exact bf16 pieces of dsm_split3, packed as MFMA operands (element j in half-word j)
__device__ __forceinline__ void dsm_split3_pack8(const float4& a, const float4& b, dsm_bf16x8& lo, dsm_bf16x8& mid, dsm_bf16x8& hi) {
  const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  uint32_t uh[8], um[8], ul[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const uint32_t u = __float_as_uint(x[j]);
    const float r1 = x[j] - __uint_as_float(u & 0xFFFF0000u);
    const uint32_t u1 = __float_as_uint(r1);
    const float r2 = r1 - __uint_as_float(u1 & 0xFFFF0000u);
    uh[j] = u; um[j] = u1; ul[j] = __float_as_uint(r2);
  }
  dsm_u32x4v ph, pm, pl;
#pragma unroll
  for (int p = 0; p < 4; ++p) {  // the high half-words of a pair, element 2p in the low half
    ph[p] = __builtin_amdgcn_perm(uh[2 * p + 1], uh[2 * p], 0x07060302u);
    pm[p] = __builtin_amdgcn_perm(um[2 * p + 1], um[2 * p], 0x07060302u);
    pl[p] = __builtin_amdgcn_perm(ul[2 * p + 1], ul[2 * p], 0x07060302u);
  }
  hi = __builtin_bit_cast(dsm_bf16x8, ph);
  mid = __builtin_bit_cast(dsm_bf16x8, pm);
  lo = __builtin_bit_cast(dsm_bf16x8, pl);
}

// NW waves per workgroup, wave w owns chunks [w CPW, (w+1) CPW); DX = depth of the activation window in 32-wide blocks.
// grid (n tiles, 1, m tiles).  Dynamic LDS: chunks x NT MT x 1 KB.
template <typename KVT, int MT, int NT, int EPI, int CPW, int NW, int OCC, bool BX3, int DX = 2>
__global__ __launch_bounds__(64 * NW, OCC) void gemm_wk_kernel(GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) float wk_part[];
  launch_stamp_begin(a.ts);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  constexpr int TILES = NT * MT;
  constexpr int NB = 8 * CPW;  // 32-wide blocks per wave
  const int m_base = blockIdx.z * (16 * MT);
  const int n_base = blockIdx.x * ((EPI == EPI_GATE) ? 16 : 16 * NT);
  const int nblk_all = a.Kpad >> 5;
  const int chunks = (a.Kpad + DSM_KC - 1) / DSM_KC;
  const int kb0 = wave * NB;                       // this wave's first block
  const int nkb = min(NB, nblk_all - kb0);         // its block count (<= 0: a wave without work)
  const uint16_t* W = reinterpret_cast<const uint16_t*>(a.W);

  f32x4 acc[NT][MT];
  if (nkb > 0) {
    const uint16_t* wrow[NT];
    const int wst = dsm_wstep<uint16_t>(a);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) wrow[nt] = dsm_wbase<uint16_t>(a, W, n_base + nt * a.nt_stride, r, q) + (long)32 * kb0 * wst;
    const float* xrow[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      int m = m_base + 16 * mt + r;
      m = m < a.M ? m : a.M - 1;  // padded rows re-read the last real row; their results are discarded
      xrow[mt] = a.X + a.xmap.off(m) + 8 * q + 32 * kb0;
    }
    // activations of the first DX blocks, then every weight fragment of the wave (block indices clamped to the last real
    // block: loaded again, never used — no branch around a load)
    float4 xv[DX][MT][2];
#pragma unroll
    for (int i = 0; i < DX; ++i) {
      const int kb = 32 * (i < nkb ? i : nkb - 1);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        xv[i][mt][0] = *reinterpret_cast<const float4*>(xrow[mt] + kb);
        xv[i][mt][1] = *reinterpret_cast<const float4*>(xrow[mt] + kb + 4);
      }
    }
    uint4 wv[NB][NT];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int kb = 32 * (i < nkb ? i : nkb - 1);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) wv[i][nt] = *reinterpret_cast<const uint4*>(wrow[nt] + (long)kb * wst);
    }
    __builtin_amdgcn_sched_barrier(0);  // every request is out before the first block waits for its own
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      if (i < nkb) {  // wave-uniform
        float4 xc[MT][2];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) { xc[mt][0] = xv[i % DX][mt][0]; xc[mt][1] = xv[i % DX][mt][1]; }
        if (i + DX < NB) {  // the slot is free again: DX blocks ahead
          const int kb = 32 * (i + DX < nkb ? i + DX : nkb - 1);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            xv[i % DX][mt][0] = *reinterpret_cast<const float4*>(xrow[mt] + kb);
            xv[i % DX][mt][1] = *reinterpret_cast<const float4*>(xrow[mt] + kb + 4);
          }
        }
        if (BX3) {
          dsm_bf16x8 wa[NT];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const dsm_u32x4v t = {wv[i][nt].x, wv[i][nt].y, wv[i][nt].z, wv[i][nt].w};
            wa[nt] = __builtin_bit_cast(dsm_bf16x8, t);
          }
          dsm_bf16x8 xp[3][MT];
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) dsm_split3_pack8(xc[mt][0], xc[mt][1], xp[0][mt], xp[1][mt], xp[2][mt]);
#pragma unroll
          for (int p = 0; p < 3; ++p)  // canonical order of the pieces: lo, mid, hi
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
              for (int nt = 0; nt < NT; ++nt) acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[nt], xp[p][mt], acc[nt][mt], 0, 0, 0);
        } else {
          float wa[NT][8];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const uint4 v = wv[i][nt];
            wa[nt][0] = __uint_as_float(v.x << 16); wa[nt][1] = __uint_as_float(v.x & 0xFFFF0000u);
            wa[nt][2] = __uint_as_float(v.y << 16); wa[nt][3] = __uint_as_float(v.y & 0xFFFF0000u);
            wa[nt][4] = __uint_as_float(v.z << 16); wa[nt][5] = __uint_as_float(v.z & 0xFFFF0000u);
            wa[nt][6] = __uint_as_float(v.w << 16); wa[nt][7] = __uint_as_float(v.w & 0xFFFF0000u);
          }
#pragma unroll
          for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
              for (int mt = 0; mt < MT; ++mt) {
                const float xs = s < 4 ? (s == 0 ? xc[mt][0].x : s == 1 ? xc[mt][0].y : s == 2 ? xc[mt][0].z : xc[mt][0].w)
                                       : (s == 4 ? xc[mt][1].x : s == 5 ? xc[mt][1].y : s == 6 ? xc[mt][1].z : xc[mt][1].w);
                acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[nt][s], xs, acc[nt][mt], 0, 0, 0);
              }
        }
        if ((i & 7) == 7 || i == nkb - 1) {  // a 256-wide chunk is complete: park its sum
          const int chunk = (kb0 + i) >> 3;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
              *reinterpret_cast<f32x4*>(&wk_part[((chunk * TILES + nt * MT + mt) * 64 + lane) * 4]) = acc[nt][mt];
              acc[nt][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
      }
    }
  }
  __syncthreads();
  // canonical split-K order: +0 + c0 + c1 + ... left to right, by the wave that owns the output tile.  EPI_GATE: the owner of
  // m-tile mt needs the gate and the up tile; otherwise tile (nt, mt) belongs to wave (nt MT + mt) mod NW.
  constexpr int OUT = (EPI == EPI_GATE) ? MT : TILES;
#pragma unroll
  for (int t = 0; t < OUT; ++t) {
    if ((t % NW) != wave) continue;  // wave-uniform
    if (EPI == EPI_GATE) {
      f32x4 g = (f32x4){0.f, 0.f, 0.f, 0.f}, u = (f32x4){0.f, 0.f, 0.f, 0.f};
      for (int c = 0; c < chunks; ++c) {
        g = g + *reinterpret_cast<const f32x4*>(&wk_part[((c * TILES + t) * 64 + lane) * 4]);
        u = u + *reinterpret_cast<const f32x4*>(&wk_part[((c * TILES + (NT - 1) * MT + t) * 64 + lane) * 4]);
      }
      epi_gate(a, g, u, m_base + 16 * t + r, n_base + 4 * q);
    } else {
      const int nt = t / MT, mt = t % MT;
      f32x4 tot = (f32x4){0.f, 0.f, 0.f, 0.f};
      for (int c = 0; c < chunks; ++c) tot = tot + *reinterpret_cast<const f32x4*>(&wk_part[((c * TILES + t) * 64 + lane) * 4]);
      const int m = m_base + 16 * mt + r, n = n_base + nt * a.nt_stride + 4 * q;
      if (EPI == EPI_RVQ)
        epi_rvq(a, tot, m, n, (n_base + nt * a.nt_stride) >> 4, q);
      else
        epi_store_qkv<KVT, EPI>(a, tot, m, n);
    }
  }
  launch_stamp_end(a.ts);
}

// ---- dot_mode 1, split-K form with every load of the chunk issued up front (r04) ------------------------------------------
// Same tile, grid, slabs and epilogues as gemm_bx3_kernel<.., LOOP = false> (64 NT weight rows x 16 MT activation rows x one
// 256-wide chunk per workgroup) and the same arithmetic, but where that kernel fetches block g + 1 while it multiplies block g
// — eight dependent HBM round trips per wave — this one requests the whole chunk before it waits for anything: first the
// activations (4 MT 16-byte pieces per thread, L2 hits, so they land first and vmcnt — which retires in issue order — lets the
// split start while the weights are still in flight), then the 8 NT weight fragments.  The activations are split into their
// three bf16 planes and staged for HB blocks at a time (HB = 8: the whole chunk behind ONE barrier, 24 MT KB of LDS; HB = 4:
// two halves, 12 MT KB), then the blocks are multiplied back to back, each waiting only for its own weight fragment.
// WL: weight layout.  0: row-major [N][Kpad]; 1: tile-major [N / 16][Kpad / 32][64 lanes][8] — the 16 x 32 fragment one wave
// loads per block is 1 KB contiguous, a wave's chunk 8 KB; 2: chunk-major [chunk][N / 16][blocks of the chunk][64][8] — the 64
// rows x 256 k a workgroup streams are 32 KB contiguous and consecutive workgroups follow each other in memory.
template <typename KVT, int MT, int NT, int EPI, int HB, int WL = 0, int LATE = 0>
__global__ __launch_bounds__(256, LATE ? 4 : 2) void gemm_bx3u_kernel(GemmArgs a) {
  // DYNAMIC LDS (HB x 3 x 16 MT x 32 bf16 = 48 KB at HB = 8, MT = 2), not a static array: with a static 48 KB the compiler knows
  // that only three workgroups fit on a CU and pads the kernel's register allocation up to that occupancy (136 VGPRs allocated for
  // 88-100 used) — registers another stream's workgroups could have had beside it.
  extern __shared__ __attribute__((aligned(16))) uint16_t Xp[];  // [HB][3][16 MT][32]
  constexpr int XP_BLK = 3 * 16 * MT * 32, XP_PLANE = 16 * MT * 32;
  launch_stamp_begin(a.ts);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef DSM_WK_STAMPS
  unsigned long long* stamp = a.ts ? a.ts + 16 + 8 * (size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) : nullptr;
  if (stamp && tid == 0) stamp[6] = wall_clock64();
#endif
  const int r = lane & 15, q = lane >> 4;
  const int chunks = (int)gridDim.y;
  const int m_base = blockIdx.z * (16 * MT);
  const int n_base = blockIdx.x * 64 + 16 * wave;
  const int kb0 = (int)blockIdx.y * 8;
  const int nkb = min(8, (a.Kpad >> 5) - kb0);  // 1..8 blocks in this chunk
  const uint16_t* W = reinterpret_cast<const uint16_t*>(a.W);
  // activation pieces: the chunk of a row is 64 float4; piece p = tid + 256 i -> row p / 64, float4 p % 64 (a wave reads one
  // KB of one row per instruction).  Pieces beyond a partial last chunk re-read the chunk's last block and are not staged.
  constexpr int NP = 4 * MT;
  float4 xv[NP];
  const int c4 = tid & 63;  // float4 index inside the chunk row: block c4 / 8, part c4 % 8 — the same for every piece of a thread
  const int xblk = c4 >> 3, part = c4 & 7;
  const int xoff = 32 * (kb0 + (xblk < nkb ? xblk : nkb - 1)) + 4 * part;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    int m = m_base + (tid >> 6) + 4 * i;
    m = m < a.M ? m : a.M - 1;
    xv[i] = *reinterpret_cast<const float4*>(a.X + a.xmap.off(m) + xoff);
  }
  // LATE > 0 (r04, late): the last LATE blocks' weight fragments are requested only after the activations have been split and
  // staged — their registers are the ones the activation pieces held, so the kernel fits 128 VGPRs (four waves per SIMD: two of
  // its workgroups fit on a CU beside two attention workgroups of the other stream group instead of one)
  static_assert(LATE == 0 || HB == 8, "the late half is requested between the staging and its barrier");
  uint4 wv[8][NT];
#define DSM_WLOAD(G0, G1)                                                                                                              \
  _Pragma("unroll") for (int g = (G0); g < (G1); ++g) {                                                                               \
    const int kb = 32 * (kb0 + (g < nkb ? g : nkb - 1));                                                                               \
    _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) {                                                                                \
      const int gc = g < nkb ? g : nkb - 1;                                                                                            \
      if (WL == 0)                                                                                                                     \
        wv[g][nt] = *reinterpret_cast<const uint4*>(dsm_wbase<uint16_t>(a, W, n_base + nt * a.nt_stride, r, q) + (long)kb * dsm_wstep<uint16_t>(a)); \
      else if (WL == 1)                                                                                                                \
        wv[g][nt] = *reinterpret_cast<const uint4*>(W + ((long)((n_base + nt * a.nt_stride) >> 4) * (a.Kpad >> 5) + kb0 + gc) * 512 + lane * 8); \
      else                                                                                                                             \
        wv[g][nt] = *reinterpret_cast<const uint4*>(W + (((long)blockIdx.y * a.w_ntiles * 8) + (long)((n_base + nt * a.nt_stride) >> 4) * nkb + gc) * 512 + lane * 8); \
    }                                                                                                                                  \
  }
  DSM_WLOAD(0, 8 - LATE)
  __builtin_amdgcn_sched_barrier(0);
#ifdef DSM_WK_STAMPS  // experiments/gemm_wk_probe.hip: per-workgroup phase stamps (wall clock, 10 ns)
#define DSM_STAMP(i) if (stamp && tid == 0) stamp[i] = wall_clock64();
#define DSM_STAMP_WAIT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n));
#else
#define DSM_STAMP(i)
#define DSM_STAMP_WAIT(n)
#endif
  DSM_STAMP(0)
  f32x4 acc[NT][MT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ph = 0; ph < 8 / HB; ++ph) {
    if (ph > 0) __syncthreads();  // the previous half's fragments have been read
    if (xblk >= ph * HB && xblk < (ph + 1) * HB && xblk < nkb) {
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int row = (tid >> 6) + 4 * i;
        uint16_t h[4], m[4], l[4];
        dsm_split3(xv[i].x, &h[0], &m[0], &l[0]); dsm_split3(xv[i].y, &h[1], &m[1], &l[1]);
        dsm_split3(xv[i].z, &h[2], &m[2], &l[2]); dsm_split3(xv[i].w, &h[3], &m[3], &l[3]);
        const int off = row * 32 + (((part >> 1) ^ ((row >> 1) & 3)) * 8) + (part & 1) * 4;
        uint16_t* base = Xp + (xblk - ph * HB) * XP_BLK;
        *reinterpret_cast<uint2*>(base + off) = make_uint2((uint32_t)l[0] | ((uint32_t)l[1] << 16), (uint32_t)l[2] | ((uint32_t)l[3] << 16));
        *reinterpret_cast<uint2*>(base + 16 * MT * 32 + off) = make_uint2((uint32_t)m[0] | ((uint32_t)m[1] << 16), (uint32_t)m[2] | ((uint32_t)m[3] << 16));
        *reinterpret_cast<uint2*>(base + 2 * 16 * MT * 32 + off) = make_uint2((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16));
      }
    }
    if (LATE > 0) {
      __builtin_amdgcn_sched_barrier(0);  // the staging stores above, then the late requests: not the other way round
      DSM_WLOAD(8 - LATE, 8)
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    if (ph == 0) { DSM_STAMP(1) }
#pragma unroll
    for (int gg = 0; gg < HB; ++gg) {
      const int g = ph * HB + gg;
      if (g == 0) { DSM_STAMP_WAIT(8 * NT - NT) DSM_STAMP(2) }
      if (g == 7) { DSM_STAMP_WAIT(0) DSM_STAMP(3) }
      if (g < nkb) {  // workgroup-uniform
        dsm_bf16x8 wa[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const dsm_u32x4v t = {wv[g][nt].x, wv[g][nt].y, wv[g][nt].z, wv[g][nt].w};
          wa[nt] = __builtin_bit_cast(dsm_bf16x8, t);
        }
#pragma unroll
        for (int p = 0; p < 3; ++p)  // canonical order of the pieces: lo, mid, hi
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            const int row = 16 * mt + r;
            const dsm_bf16x8 xf = *reinterpret_cast<const dsm_bf16x8*>(Xp + gg * XP_BLK + p * XP_PLANE + row * 32 + ((q ^ ((row >> 1) & 3)) * 8));
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[nt], xf, acc[nt][mt], 0, 0, 0);
          }
      }
    }
  }
  DSM_STAMP(4)
  f32x4 tot[NT][MT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) tot[nt][mt] = (f32x4){0.f, 0.f, 0.f, 0.f} + acc[nt][mt];  // +0 + chunk sum, as gemm_bx3_kernel
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
    DSM_STAMP_WAIT(0)
    DSM_STAMP(5)
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
#undef DSM_STAMP
#undef DSM_STAMP_WAIT
#undef DSM_WLOAD
}

// ---- gemm_wk_kernel with the row norm of its input in the prologue (r04) -----------------------------------------------------
// The GEMM that follows a norm reads the whole rows anyway when the whole K lives in the workgroup — so the norm launch (and the
// xn round trip) goes: X is the raw residual stream, every workgroup computes the statistics of ITS 16 rows in the canonical
// order of row_norm_kernel (dsm_numerics.h, ROW SUM) and normalises on the way into the B operand.  d = K <= 1024: thread t of
// the canonical 256 owns elements 4t .. 4t+3, so canonical wave w IS this kernel's wave w (chunk w = columns 256 w ..), and a
// lane (r, q) holds, for block i and half e, the four elements of canonical lane L = 8 i + 2 q + e of row r: the xor butterfly
// 32, 16, 8, 4, 2, 1 over L is a sum over i (bits 2, 1, 0 of i: in registers), q (lanes ^ 32, ^ 16) and e (in registers), in that
// order — a + b is commutative, so the tree gives every lane the bits the butterfly gives; wave totals meet in LDS and are added
// left to right.  Then (x / mm) * w, or ((x - mean) * inv) * w + b: the arithmetic of row_norm_apply.
// Four waves, one chunk each (a wave beyond the last chunk contributes +0 totals), 16 rows x 16 NT columns per workgroup,
// dot_mode 1 only.  Every load — activations, norm weights, weight fragments — is issued before the first wait.
// The norm weights (and LayerNorm bias) of a wave's chunk — 256 floats, the same for its 16 rows — travel as one float4 per lane
// into wave-private LDS and are read back as broadcasts: 4 registers instead of 64.
template <typename KVT, int NT, int EPI, bool RMS>
__global__ __launch_bounds__(256, 2) void gemm_wkn_kernel(GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) float wk_part[];  // [chunks][NT][64][4] partial tiles, [2][4][16] wave totals, [2][4][256] norm w / b
  launch_stamp_begin(a.ts);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int m_base = blockIdx.z * 16;
  const int n_base = blockIdx.x * ((EPI == EPI_GATE) ? 16 : 16 * NT);
  const int chunks = (a.Kpad + DSM_KC - 1) / DSM_KC;  // <= 4
  const int kb0 = wave * 8;
  const int nkb = min(8, (a.Kpad >> 5) - kb0);  // <= 0: no chunk for this wave
  float* red = wk_part + chunks * NT * 256;
  float* nws = red + 128 + wave * 256;
  float* nbs = red + 128 + 1024 + wave * 256;
  const uint16_t* W = reinterpret_cast<const uint16_t*>(a.W);
  const int d = a.K;
  float4 xv[8][2];
  uint4 wv[8][NT];
  if (nkb > 0) {
    int m = m_base + r;
    m = m < a.M ? m : a.M - 1;
    const float* xrow = a.X + a.xmap.off(m) + 8 * q + 32 * kb0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int kb = 32 * (i < nkb ? i : nkb - 1);
      xv[i][0] = *reinterpret_cast<const float4*>(xrow + kb);
      xv[i][1] = *reinterpret_cast<const float4*>(xrow + kb + 4);
    }
    {
      int k = 32 * kb0 + 4 * lane;
      k = k + 4 <= d ? k : d - 4;  // a partial last chunk: the tail lanes re-read the last four weights, which no block uses
      const float4 w4 = *reinterpret_cast<const float4*>(a.pre_norm_w + k);
      *reinterpret_cast<float4*>(nws + 4 * lane) = w4;
      if (!RMS) *reinterpret_cast<float4*>(nbs + 4 * lane) = *reinterpret_cast<const float4*>(a.pre_norm_b + k);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int kb = 32 * (i < nkb ? i : nkb - 1);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        wv[i][nt] = *reinterpret_cast<const uint4*>(dsm_wbase<uint16_t>(a, W, n_base + nt * a.nt_stride, r, q) + (long)(32 * kb0 + kb) * dsm_wstep<uint16_t>(a));
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  // canonical lane (i, q, e): chain over its four elements from +0
  float s = 0.0f, s2 = 0.0f;
  {
    float ps[8][2], ps2[8][2];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        float cs = 0.0f, cs2 = 0.0f;
        if (i < nkb) {
          const float4 v = xv[i][e];
          cs = cs + v.x; cs2 = DSM_FMAF(v.x, v.x, cs2);
          cs = cs + v.y; cs2 = DSM_FMAF(v.y, v.y, cs2);
          cs = cs + v.z; cs2 = DSM_FMAF(v.z, v.z, cs2);
          cs = cs + v.w; cs2 = DSM_FMAF(v.w, v.w, cs2);
        }
        ps[i][e] = cs; ps2[i][e] = cs2;
      }
    // butterfly offsets 32, 16, 8 of L: bits 2, 1, 0 of the block index
#pragma unroll
    for (int e = 0; e < 2; ++e) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { ps[i][e] = ps[i][e] + ps[i + 4][e]; ps2[i][e] = ps2[i][e] + ps2[i + 4][e]; }
#pragma unroll
      for (int i = 0; i < 2; ++i) { ps[i][e] = ps[i][e] + ps[i + 2][e]; ps2[i][e] = ps2[i][e] + ps2[i + 2][e]; }
      ps[0][e] = ps[0][e] + ps[1][e]; ps2[0][e] = ps2[0][e] + ps2[1][e];
      // offsets 4, 2: the two bits of q = lane bits 5, 4
      ps[0][e] = ps[0][e] + __shfl_xor(ps[0][e], 32, 64); ps2[0][e] = ps2[0][e] + __shfl_xor(ps2[0][e], 32, 64);
      ps[0][e] = ps[0][e] + __shfl_xor(ps[0][e], 16, 64); ps2[0][e] = ps2[0][e] + __shfl_xor(ps2[0][e], 16, 64);
    }
    s = ps[0][0] + ps[0][1];  // offset 1
    s2 = ps2[0][0] + ps2[0][1];
  }
  if (q == 0) { red[wave * 16 + r] = s; red[64 + wave * 16 + r] = s2; }
  __syncthreads();
  s = ((red[r] + red[16 + r]) + red[32 + r]) + red[48 + r];
  s2 = ((red[64 + r] + red[80 + r]) + red[96 + r]) + red[112 + r];
  float mm = 1.f, mean = 0.f, inv = 0.f;
  if (RMS) {
    mm = sqrtf(s2 / (float)d + a.pre_norm_eps);
  } else {
    mean = s / (float)d;
    const float var = s2 / (float)d - mean * mean;
    inv = 1.0f / sqrtf(var + a.pre_norm_eps);
  }
  f32x4 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (nkb > 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (i < nkb) {  // wave-uniform
        float4 xn[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const float4 v = xv[i][e], w4 = *reinterpret_cast<const float4*>(nws + 32 * i + 8 * q + 4 * e);
          if (RMS) {
            xn[e].x = (v.x / mm) * w4.x; xn[e].y = (v.y / mm) * w4.y; xn[e].z = (v.z / mm) * w4.z; xn[e].w = (v.w / mm) * w4.w;
          } else {
            const float4 b4 = *reinterpret_cast<const float4*>(nbs + 32 * i + 8 * q + 4 * e);
            xn[e].x = ((v.x - mean) * inv) * w4.x + b4.x; xn[e].y = ((v.y - mean) * inv) * w4.y + b4.y;
            xn[e].z = ((v.z - mean) * inv) * w4.z + b4.z; xn[e].w = ((v.w - mean) * inv) * w4.w + b4.w;
          }
        }
        dsm_bf16x8 xp[3];
        dsm_split3_pack8(xn[0], xn[1], xp[0], xp[1], xp[2]);
        dsm_bf16x8 wa[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const dsm_u32x4v t = {wv[i][nt].x, wv[i][nt].y, wv[i][nt].z, wv[i][nt].w};
          wa[nt] = __builtin_bit_cast(dsm_bf16x8, t);
        }
#pragma unroll
        for (int p = 0; p < 3; ++p)  // lo, mid, hi
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[nt], xp[p], acc[nt], 0, 0, 0);
      }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) *reinterpret_cast<f32x4*>(&wk_part[((wave * NT + nt) * 64 + lane) * 4]) = acc[nt];
  }
  __syncthreads();
  constexpr int OUT = (EPI == EPI_GATE) ? 1 : NT;
#pragma unroll
  for (int t = 0; t < OUT; ++t) {
    if ((t % 4) != wave) continue;
    if (EPI == EPI_GATE) {
      f32x4 g = (f32x4){0.f, 0.f, 0.f, 0.f}, u = (f32x4){0.f, 0.f, 0.f, 0.f};
      for (int c = 0; c < chunks; ++c) {
        g = g + *reinterpret_cast<const f32x4*>(&wk_part[((c * NT) * 64 + lane) * 4]);
        u = u + *reinterpret_cast<const f32x4*>(&wk_part[((c * NT + NT - 1) * 64 + lane) * 4]);
      }
      epi_gate(a, g, u, m_base + r, n_base + 4 * q);
    } else {
      f32x4 tot = (f32x4){0.f, 0.f, 0.f, 0.f};
      for (int c = 0; c < chunks; ++c) tot = tot + *reinterpret_cast<const f32x4*>(&wk_part[((c * NT + t) * 64 + lane) * 4]);
      epi_store_qkv<KVT, EPI>(a, tot, m_base + r, n_base + t * a.nt_stride + 4 * q);
    }
  }
  launch_stamp_end(a.ts);
}
