// ll_dataflow_probe.hip — a stage boundary WITHOUT a barrier: every activation word travels as an 8-byte packet {value, tag}
// (tag = the stage that produced it; an aligned 8-byte store is single-copy atomic), consumers poll the packets they need until
// the tags say "this stage" — the low-latency protocol of collective libraries applied to a chain of small GEMM stages.  (r04)
// grid_barrier_probe2 showed that the barrier form costs what a launch costs: 2.2 us for the barrier alone, ~6 us with a store
// acknowledgement in front and an uncached read behind it.  Here the only serial cost is one store -> visible -> load latency.
// Shape of one stage (the DepFormer's out_proj at 32 rows): activations [32][1024], 128 workgroups = 2 row tiles x 64 column
// tiles; each reads its row tile's 16 x 1024 packets (128 KB: wave w the 256-wide chunk w, lane (r, q) eight consecutive packets
// per 32-wide block — the MFMA B-operand pattern of gemm_wk_kernel), checks every tag and every value, and writes 16 x 16 packets
// of the next activation matrix (ping-pong buffers).  SPREAD > 1: only every SPREAD-th stage is "wide" like this; the others
// have 512 units of 16 x 64 inputs (attention-like), to see the cost of a small stage.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP %s @%d\n", hipGetErrorString(e_), __LINE__); exit(2);} } while (0)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// two relaxed agent-scope 8-byte atomic loads (global_load_dwordx2 sc1): the compiler places the s_waitcnt itself.  (An
// asynchronous load in inline assembly with an "=v" output is a trap: the compiler may move the register before the wait.)
__device__ __forceinline__ u32x4 ld16(const void* p) {
  const unsigned long long a = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long b = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return (u32x4){(unsigned)a, (unsigned)(a >> 32), (unsigned)b, (unsigned)(b >> 32)};
}
// s_nop: a 16-byte store reads its data registers a cycle after it issues; the compiler's hazard recognizer cannot see inside
// inline assembly, and without the wait state the next store's operands overwrote this one's in lanes 12, 28, 44, 60 (found by
// this probe's own value check)
__device__ __forceinline__ void st16(void* p, u32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void wait_loads() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__device__ __forceinline__ float pattern(int stage, int row, int col) { return (float)((stage * 7 + row * 13 + col) & 1023); }

struct Pkt { float v; unsigned tag; };

// bufs: 2 x [32][1024] packets.  Stage s reads bufs[(s-1)&1] (tags s-1) and writes bufs[s&1] (tags s).
template <int NBLK>  // 32-wide blocks per wave that are really loaded (8 = the whole chunk)
__global__ __launch_bounds__(256) void chain(Pkt* bufs, int iters, unsigned long long* out, unsigned* abort_word) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
  const int rt = blockIdx.x & 1, ct = blockIdx.x >> 1;
  unsigned long long bad = 0, polls = 0;
  bool dead = false;
  const unsigned long long t0 = wall_clock64();
  for (int s = 1; s <= iters && !dead; ++s) {
    const Pkt* X = bufs + (size_t)((s - 1) & 1) * 32 * 1024;
    Pkt* Y = bufs + (size_t)(s & 1) * 32 * 1024;
    const Pkt* xrow = X + (size_t)(16 * rt + r) * 1024 + 256 * wave + 8 * q;
    u32x4 pk[NBLK][4];
    int spins = 0;
    for (;;) {
#pragma unroll
      for (int i = 0; i < NBLK; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) pk[i][j] = ld16(xrow + 32 * i + 2 * j);
      bool ok = true;
#pragma unroll
      for (int i = 0; i < NBLK; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) ok = ok && (int)(pk[i][j][1] - (unsigned)(s - 1)) >= 0 && (int)(pk[i][j][3] - (unsigned)(s - 1)) >= 0;  // a later tag only when NBLK < 8 lets workgroups run ahead
      if (__all(ok)) break;
      ++polls;
      if (++spins > (1 << 20) || __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { dead = true; break; }
      __builtin_amdgcn_s_sleep(1);
    }
    if (dead) { __hip_atomic_store(abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    // every wave must agree before anyone writes (a dead wave must not leave the others waiting in a barrier: all leave)
    __shared__ int dead_s;
    if (tid == 0) dead_s = 0;
    __syncthreads();
    if (dead) dead_s = 1;
    __syncthreads();
    if (dead_s) { dead = true; break; }
    if (s > 1) {
#pragma unroll
      for (int i = 0; i < NBLK; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int col = 256 * wave + 32 * i + 8 * q + 2 * j;
          const bool b0 = pk[i][j][1] == (unsigned)(s - 1) && __uint_as_float(pk[i][j][0]) != pattern(s - 1, 16 * rt + r, col);
          const bool b1 = pk[i][j][3] == (unsigned)(s - 1) && __uint_as_float(pk[i][j][2]) != pattern(s - 1, 16 * rt + r, col + 1);
          if ((b0 || b1) && atomicAdd(out + 3, 1ull) == 0) {  // the first mismatch, for the record
            out[4] = (unsigned long long)s; out[5] = (unsigned long long)(16 * rt + r); out[6] = (unsigned long long)(col + (b0 ? 0 : 1));
            out[7] = ((unsigned long long)pk[i][j][b0 ? 0 : 2] << 32) | pk[i][j][b0 ? 1 : 3];
          }
          bad += b0; bad += b1;
        }
    }
    if (wave == 0) {  // 16 x 16 outputs: lane (r, q) row r, columns 4q .. 4q+3
      const int row = 16 * rt + r, col = 16 * ct + 4 * q;
      Pkt* y = Y + (size_t)row * 1024 + col;
      st16(y, (u32x4){__float_as_uint(pattern(s, row, col)), (unsigned)s, __float_as_uint(pattern(s, row, col + 1)), (unsigned)s});
      st16(y + 2, (u32x4){__float_as_uint(pattern(s, row, col + 2)), (unsigned)s, __float_as_uint(pattern(s, row, col + 3)), (unsigned)s});
    }
  }
  const unsigned long long t1 = wall_clock64();
  atomicAdd(out + 1, bad);
  atomicAdd(out + 2, polls);
  if (blockIdx.x == 0 && tid == 0) out[0] = t1 - t0;
}

template <int NBLK>
static void run(int uncached, int iters) {
  Pkt* bufs;
  const size_t bytes = (size_t)2 * 32 * 1024 * sizeof(Pkt);
  if (uncached) CK(hipExtMallocWithFlags((void**)&bufs, bytes, hipDeviceMallocUncached)); else CK(hipMalloc(&bufs, bytes));
  unsigned long long* out; CK(hipMalloc(&out, 64));
  unsigned* ab; CK(hipMalloc(&ab, 64));
  CK(hipMemset(bufs, 0, bytes)); CK(hipMemset(out, 0, 64)); CK(hipMemset(ab, 0, 64));  // tag 0 = "stage 0" everywhere: stage 1 starts at once
  CK(hipDeviceSynchronize());
  hipLaunchKernelGGL((chain<NBLK>), dim3(128), dim3(256), 0, 0, bufs, iters, out, ab);
  CK(hipDeviceSynchronize());
  unsigned long long h[8]; unsigned a;
  CK(hipMemcpy(h, out, 64, hipMemcpyDeviceToHost)); CK(hipMemcpy(&a, ab, 4, hipMemcpyDeviceToHost));
  printf("%s, %d of 8 blocks per wave read (%3d KB per workgroup): %.2f us per stage, %llu value mismatches, %.1f re-polls per wave and stage%s\n",
         uncached ? "uncached alloc" : "hipMalloc     ", NBLK, NBLK * 16, (double)h[0] * 10.0 / 1000.0 / iters, h[1],
         (double)h[2] / (128.0 * 4 * 64) / iters, a ? "  ABORTED (spin bound)" : "");
  {  // host-side check of what the last stage left behind
    static Pkt hb[32 * 1024];
    CK(hipMemcpy(hb, bufs + (size_t)(iters & 1) * 32 * 1024, sizeof hb, hipMemcpyDeviceToHost));
    long wrong = 0; int fr = -1, fc = -1;
    for (int r = 0; r < 32; ++r) for (int c = 0; c < 1024; ++c) {
      const bool w = hb[r * 1024 + c].tag != (unsigned)iters || hb[r * 1024 + c].v != (float)((iters * 7 + r * 13 + c) & 1023);
      if (w && fr < 0) { fr = r; fc = c; }
      wrong += w;
    }
    printf("    host check of the last stage's output: %ld of 32768 packets wrong", wrong);
    if (wrong) printf(" (first: row %d col %d holds %g tag %u, expected %g)", fr, fc, (double)hb[fr * 1024 + fc].v, hb[fr * 1024 + fc].tag, (double)((iters * 7 + fr * 13 + fc) & 1023));
    printf("\n");
  }
  if (h[1]) printf("    first mismatch: stage %llu row %llu col %llu: value %g tag %llu, expected %g\n", h[4], h[5], h[6], (double)__builtin_bit_cast(float, (unsigned)(h[7] >> 32)), h[7] & 0xFFFFFFFFull, (double)(float)(((int)h[4] - 1) * 7 + (int)h[5] * 13 + (int)h[6] & 1023));
  fflush(stdout);
  CK(hipFree(bufs)); CK(hipFree(out)); CK(hipFree(ab));
}

int main() {
  const int iters = 2000;
  for (int unc : {1, 0}) {
    run<8>(unc, iters);
    run<4>(unc, iters);
    run<2>(unc, iters);
    run<1>(unc, iters);
  }
  return 0;
}
