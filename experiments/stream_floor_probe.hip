// stream_floor_probe.hip — how fast can ONE short launch pull its weights, whatever it then does with them? (r04)
// The LM GEMMs at 32 rows are launches of 700-770 workgroups that read 8-46 MB once.  gemm_bx3u_kernel issues every load of a
// workgroup up front into registers and measures 14.2 / 9.1 / 8.6 / 5.4 us for gate / QKV / ff_out / out_proj.  This probe takes the
// arithmetic away: the same grids, the same bytes per workgroup, three ways of asking for them —
//   R: 16-byte loads into registers, all issued before the first use (what the kernels do)
//   N: the same, non-temporal
//   L: global_load_lds_dwordx4 straight into LDS (no VGPRs for the data), one s_waitcnt at the end
// — and a store that (almost) never happens but depends on every lane's data, so that nothing is optimised away.  Back-to-back launches over ROT distinct copies of the buffer
// (more than the 256 MB Infinity Cache), hipEvent-timed over many launches.  The difference to the GEMM times is what staging,
// MFMA and the slab stores cost; the R time is the floor of the current structure.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP %s @%d\n", hipGetErrorString(e_), __LINE__); exit(2);} } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NL, int MODE>  // NL 16-byte loads per thread
__global__ __launch_bounds__(256, 2) void stream_kernel(const u32x4* __restrict__ w, u32x4* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const size_t base = ((size_t)blockIdx.x * NL) * 256 + threadIdx.x;  // each load instruction of a wave: 1 KB contiguous
  u32x4 acc = {0u, 0u, 0u, 0u};
  if (MODE == 2) {
#pragma unroll
    for (int i = 0; i < NL; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w + base + (size_t)i * 256), (__attribute__((address_space(3))) void*)(lds + ((size_t)i * 256 + (threadIdx.x & ~63)) * 16), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NL; ++i) acc ^= *reinterpret_cast<const u32x4*>(lds + ((size_t)i * 256 + threadIdx.x) * 16);
  } else {
    u32x4 v[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) v[i] = MODE == 1 ? __builtin_nontemporal_load(w + base + (size_t)i * 256) : w[base + (size_t)i * 256];
#pragma unroll
    for (int i = 0; i < NL; ++i) acc ^= v[i];
  }
  // every lane's value decides (a store by lane 0 only let the compiler sink the loads under the lane-0 branch: 1/64 of the traffic)
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x9E3779B9u) out[(blockIdx.x * 256 + threadIdx.x) & 65535] = acc;
}

// the split-K GEMM's own address pattern on fragment-major weights [N / 16][K / 32][64 lanes][8 bf16]: workgroup (bx, by) reads,
// for its four waves' n-tiles 4 bx + w (and, NT = 2, the tile `up` tiles further on), the eight 1 KB fragments of chunk by.
// ORDER 0: bx = id % gx (the kernels' launch order: concurrently running workgroups are 4 n-tiles = 4 x (K / 32) KB apart);
// ORDER 1: by = id % chunks (eight consecutive workgroups cover 4 x 64 KB contiguous pieces).
// XL: 16-byte activation loads per thread issued BEFORE the weights — the chunk's [32 rows][256] f32 block (32 KB, piece p = tid +
// 256 i as in gemm_bx3u_kernel), from a 256 KB buffer every workgroup shares (L2 hits after the first touch); LDSB: bytes of static
// LDS per workgroup (the kernels hold 48 KB: three workgroups per CU)
template <int NT, int ORDER, int XL, int LDSB>
__global__ __launch_bounds__(256, 2) void gemm_pattern_kernel(const u32x4* __restrict__ w, u32x4* __restrict__ out, int gx, int chunks, int kblocks, int up, const u32x4* __restrict__ xbuf) {
  const int id = blockIdx.x;
  const int bx = ORDER == 0 ? id % gx : id / chunks, by = ORDER == 0 ? id / gx : id % chunks;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __shared__ u32x4 pad[LDSB / 16 + 1];
  u32x4 xv[XL + 1];
#pragma unroll
  for (int i = 0; i < XL; ++i) xv[i] = xbuf[((size_t)by * 2048 + threadIdx.x + 256 * i) & 16383];
  u32x4 v[8][NT];
#pragma unroll
  for (int g = 0; g < 8; ++g)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) v[g][nt] = w[((size_t)(4 * bx + wave + nt * up) * kblocks + by * 8 + g) * 64 + lane];
  u32x4 acc = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int i = 0; i < XL; ++i) acc ^= xv[i];
  if (LDSB) { pad[threadIdx.x & (LDSB / 16 - 1)] = acc; __syncthreads(); acc ^= pad[(threadIdx.x * 7) & (LDSB / 16 - 1)]; }
#pragma unroll
  for (int g = 0; g < 8; ++g)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc ^= v[g][nt];
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x9E3779B9u) out[(blockIdx.x * 256 + threadIdx.x) & 65535] = acc;
}
template <int NT, int ORDER, int XL = 0, int LDSB = 0>
static void run_pattern(const char* name, int gx, int chunks, int up, const u32x4* w, size_t total_elems, u32x4* out) {
  const int kblocks = chunks * 8, ntiles = NT == 2 ? 2 * up : 4 * gx;
  const size_t copy = (size_t)ntiles * kblocks * 64;
  const int rot = (int)(total_elems / copy);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < rot; ++i) hipLaunchKernelGGL((gemm_pattern_kernel<NT, ORDER, XL, LDSB>), dim3(gx * chunks), dim3(256), 0, 0, w + (size_t)i * copy, out, gx, chunks, kblocks, up, w + total_elems - 16384);
  CK(hipDeviceSynchronize());
  const int reps = 4 * rot;
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((gemm_pattern_kernel<NT, ORDER, XL, LDSB>), dim3(gx * chunks), dim3(256), 0, 0, w + (size_t)(i % rot) * copy, out, gx, chunks, kblocks, up, w + total_elems - 16384);
  CK(hipEventRecord(e1, 0));
  CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1000.0 / reps, mb = (double)copy * 16 / 1e6;
  printf("  %-44s %7.2f us per launch  = %5.2f TB/s\n", name, us, mb / us);
}

template <int NL, int MODE>
static double run(const char* name, int grid, const u32x4* w, size_t copy_elems, int rot, u32x4* out) {
  const size_t lds = MODE == 2 ? (size_t)NL * 256 * 16 : 0;
  if (lds > 64 * 1024) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&stream_kernel<NL, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < rot; ++i) hipLaunchKernelGGL((stream_kernel<NL, MODE>), dim3(grid), dim3(256), lds, 0, w + (size_t)i * copy_elems, out);
  CK(hipDeviceSynchronize());
  const int reps = 4 * rot;
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((stream_kernel<NL, MODE>), dim3(grid), dim3(256), lds, 0, w + (size_t)(i % rot) * copy_elems, out);
  CK(hipEventRecord(e1, 0));
  CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1000.0 / reps, mb = (double)grid * NL * 4096 / 1e6;
  printf("  %-28s %7.2f us per launch (back to back)  = %5.2f TB/s\n", name, us, mb / us);
  return us;
}

template <int NL>
static void shape(const char* what, int grid, const u32x4* w, size_t total_elems, u32x4* out) {
  const size_t copy = (size_t)grid * NL * 256;
  const int rot = (int)(total_elems / copy);
  printf("%s: %d workgroups x %d KB = %.1f MB per launch, %d copies\n", what, grid, NL * 4, (double)copy * 16 / 1e6, rot);
  run<NL, 0>("registers", grid, w, copy, rot, out);
  run<NL, 1>("registers, non-temporal", grid, w, copy, rot, out);
  run<NL, 2>("LDS-DMA (global_load_lds x4)", grid, w, copy, rot, out);
}

__global__ void fill_hash(u32x4* w, size_t n) {  // incompressible contents (a constant fill measured 13-27 "TB/s": not DRAM traffic)
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned x = (unsigned)i * 2654435761u + 12345u;
  u32x4 v;
  for (int j = 0; j < 4; ++j) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; v[j] = x; }
  w[i] = v;
}

int main(int argc, char** argv) {
  const size_t total = (size_t)640 << 20;  // 640 MB of "weights": every shape rotates over more than the Infinity Cache
  u32x4 *w, *out;
  CK(hipMalloc(&w, total)); CK(hipMalloc(&out, 1 << 20));
  if (argc > 1 && argv[1][0] == 'c') { CK(hipMemset(w, 1, total)); printf("contents: constant bytes\n"); }
  else { hipLaunchKernelGGL(fill_hash, dim3((unsigned)((total / 16 + 255) / 256)), dim3(256), 0, 0, w, total / 16); printf("contents: hashed\n"); }
  CK(hipDeviceSynchronize());
  const size_t elems = total / 16;
  shape<16>("gate     (88 x 8 workgroups, 64 KB each)", 704, w, elems, out);
  shape<8>("QKV      (96 x 8, 32 KB each)", 768, w, elems, out);
  shape<8>("ff_out   (32 x 22, 32 KB each)", 704, w, elems, out);
  shape<8>("out_proj (32 x 8, 32 KB each)", 256, w, elems, out);
  shape<8>("dep QKV  (48 x 4 x 2, 16 KB + pad)", 384, w, elems, out);
  shape<16>("one big launch (4096 x 64 KB = 268 MB)", 4096, w, elems, out);
  printf("the GEMMs' own address pattern (fragment-major weights, registers):\n");
  run_pattern<2, 0>("gate 88 x 8, bx fastest (launch order today)", 88, 8, 352, w, elems, out);
  run_pattern<2, 1>("gate 88 x 8, chunk index fastest", 88, 8, 352, w, elems, out);
  run_pattern<2, 0, 8, 0>("gate, bx fastest + 8 activation loads", 88, 8, 352, w, elems, out);
  run_pattern<2, 0, 0, 49152>("gate, bx fastest + 48 KB LDS", 88, 8, 352, w, elems, out);
  run_pattern<2, 0, 8, 49152>("gate, bx fastest + activations + 48 KB LDS", 88, 8, 352, w, elems, out);
  run_pattern<1, 0>("QKV 96 x 8, bx fastest", 96, 8, 0, w, elems, out);
  run_pattern<1, 0, 8, 49152>("QKV, bx fastest + activations + 48 KB LDS", 96, 8, 0, w, elems, out);
  run_pattern<1, 1>("QKV 96 x 8, chunk index fastest", 96, 8, 0, w, elems, out);
  run_pattern<1, 0>("ff_out 32 x 22, bx fastest", 32, 22, 0, w, elems, out);
  run_pattern<1, 1>("ff_out 32 x 22, chunk index fastest", 32, 22, 0, w, elems, out);
  run_pattern<1, 0>("out_proj 32 x 8, bx fastest", 32, 8, 0, w, elems, out);
  run_pattern<1, 1>("out_proj 32 x 8, chunk index fastest", 32, 8, 0, w, elems, out);
  return 0;
}
