// grid_barrier_probe2.hip — what does one stage boundary of a PERSISTENT kernel cost on MI355X? (r04)
// r03's flag_latency_probe measured a point-to-point hand-off (0.6 us for the flag alone across XCDs, 1.4-2.2 us with 1-16 KB behind
// it when the payload travels with the sc0 sc1 bits and nothing is fenced; 2+ us as soon as __threadfence writes the L2 back).
// This probe measures the all-to-all form a launch-per-stage chain would be replaced by: G resident workgroups, every stage each
// of them writes W floats, passes a grid barrier and reads R floats written by 8 other workgroups in that stage, ITERS stages.
//   data path  0: hipMalloc, plain loads / stores, __threadfence() either side of the barrier (the textbook form)
//              1: hipMalloc, 16-byte loads / stores with sc0 sc1 through inline assembly, no fence
//              2: hipExtMallocWithFlags(hipDeviceMallocUncached), plain loads / stores, no fence
//              3: hipExtMallocWithFlags(hipDeviceMallocFinegrained), plain loads / stores, no fence
//   barrier    A: one flag word per workgroup (relaxed agent-scope store), wave 0 polls the whole array with one 16-byte load per lane
//              B: one counter, atomic add + poll
// Every value read is checked against what its producer must have written in that stage (mismatches are counted: a data path
// that is fast and wrong is of no use).  Spins are bounded: a workgroup that polls 2^22 times sets the abort word and every
// workgroup leaves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP %s @%d\n", hipGetErrorString(e_), __LINE__); exit(2);} } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_wt(float* p, f32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ f32x4 load_bypass(const float* p) {
  f32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ u32x4 load_flags(const unsigned* p) {
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

struct Ctl { unsigned* flags; unsigned* counter; unsigned* abort_word; };

template <int BAR>
__device__ __forceinline__ bool grid_barrier(const Ctl& c, int G, unsigned epoch) {
  __shared__ int ok_s;
  __builtin_amdgcn_s_waitcnt(0);  // this thread's stores have been acknowledged
  __syncthreads();
  if (threadIdx.x < 64) {
    bool ok = true;
    if (BAR == 0) {
      if (threadIdx.x == 0) __hip_atomic_store(c.flags + blockIdx.x, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int nl = (G + 3) / 4;
      int spins = 0;
      for (;;) {
        bool mine = true;
        if ((int)threadIdx.x < nl) {
          const u32x4 f = load_flags(c.flags + 4 * threadIdx.x);
          for (int j = 0; j < 4; ++j) if (4 * (int)threadIdx.x + j < G && (int)(f[j] - epoch) < 0) mine = false;
        }
        if (__all(mine)) break;
        if (++spins > (1 << 22) || __hip_atomic_load(c.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { ok = false; break; }
        __builtin_amdgcn_s_sleep(1);
      }
    } else {
      if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(c.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while ((int)(__hip_atomic_load(c.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - epoch * (unsigned)G) < 0) {
          if (++spins > (1 << 22) || __hip_atomic_load(c.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { ok = false; break; }
          __builtin_amdgcn_s_sleep(1);
        }
      }
      ok = __all(ok);
    }
    if (threadIdx.x == 0) {
      ok_s = ok ? 1 : 0;
      if (!ok) __hip_atomic_store(c.abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();
  return ok_s != 0;
}

__device__ __forceinline__ float pattern(int stage, int wg, int j) { return (float)((stage * 131 + wg * 7 + j) & 0xFFFF); }

// buf: 2 halves (double buffer by stage parity) x G x W floats
template <int PATH, int BAR>
__global__ __launch_bounds__(256) void stages(Ctl c, float* buf, int G, int W, int R, int iters, unsigned long long* out) {
  const int me = blockIdx.x;
  unsigned long long bad = 0;
  const unsigned long long t0 = wall_clock64();
  for (int s = 1; s <= iters; ++s) {
    float* mine = buf + ((size_t)(s & 1) * G + me) * W;
    for (int j = 4 * threadIdx.x; j < W; j += 1024) {
      const f32x4 v = {pattern(s, me, j), pattern(s, me, j + 1), pattern(s, me, j + 2), pattern(s, me, j + 3)};
      if (PATH == 1) store_wt(mine + j, v); else *reinterpret_cast<f32x4*>(mine + j) = v;
    }
    if (PATH == 0) __threadfence();
    if (!grid_barrier<BAR>(c, G, (unsigned)s)) break;
    if (PATH == 0) __threadfence();
    const int per = R / 8;  // floats from each of 8 producers
    for (int p = 0; p < 8; ++p) {
      const int src = (me + 1 + p * (G / 8 + 1)) % G;
      const float* theirs = buf + ((size_t)(s & 1) * G + src) * W;
      for (int j = 4 * threadIdx.x; j < per; j += 1024) {
        const int jj = j % W;
        const f32x4 v = PATH == 1 ? load_bypass(theirs + jj) : *reinterpret_cast<const f32x4*>(theirs + jj);
        for (int e = 0; e < 4; ++e) bad += v[e] != pattern(s, src, jj + e);
      }
    }
  }
  const unsigned long long t1 = wall_clock64();
  atomicAdd(out + 1, bad);
  if (me == 0 && threadIdx.x == 0) out[0] = t1 - t0;
}

static void* alloc_mode(int path, size_t bytes) {
  void* p = nullptr;
  if (path == 2) CK(hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached));
  else if (path == 3) CK(hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained));
  else CK(hipMalloc(&p, bytes));
  return p;
}

template <int PATH, int BAR>
static void run(int G, int W, int R, int iters) {
  float* buf = (float*)alloc_mode(PATH, (size_t)2 * G * (W > 4 ? W : 4) * 4);
  unsigned* ctl = (unsigned*)alloc_mode(PATH == 0 || PATH == 1 ? 0 : PATH, 8192);
  unsigned long long* out; CK(hipMalloc(&out, 64));
  CK(hipMemset(buf, 0, (size_t)2 * G * (W > 4 ? W : 4) * 4)); CK(hipMemset(ctl, 0, 8192)); CK(hipMemset(out, 0, 64));
  CK(hipDeviceSynchronize());
  Ctl c{ctl, ctl + 1024, ctl + 1536};
  hipLaunchKernelGGL((stages<PATH, BAR>), dim3(G), dim3(256), 0, 0, c, buf, G, W, R, iters, out);
  CK(hipDeviceSynchronize());
  unsigned long long h[2]; unsigned ab;
  CK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(&ab, ctl + 1536, 4, hipMemcpyDeviceToHost));
  static const char* pn[] = {"hipMalloc + __threadfence     ", "hipMalloc + sc0 sc1 asm       ", "Uncached alloc, plain ld/st   ", "Finegrained alloc, plain ld/st"};
  printf("G=%3d  %s  barrier %c  write %5d read %5d floats/wg: %6.2f us per stage, %llu mismatches%s\n", G, pn[PATH], BAR == 0 ? 'A' : 'B', W, R,
         (double)h[0] * 10.0 / 1000.0 / iters, h[1], ab ? "  ABORTED (spin bound)" : "");
  fflush(stdout);
  CK(hipFree(buf)); CK(hipFree(ctl)); CK(hipFree(out));
}

int main(int argc, char** argv) {
  const int iters = 500;
  for (int G : {128, 256}) {
    for (int wr = 0; wr < 3; ++wr) {
      const int W = wr == 0 ? 0 : wr == 1 ? 1024 : 4096, R = wr == 0 ? 0 : wr == 1 ? 8192 : 32768;
      run<0, 0>(G, W, R, iters);
      run<1, 0>(G, W, R, iters);
      run<2, 0>(G, W, R, iters);
      run<3, 0>(G, W, R, iters);
      run<1, 1>(G, W, R, iters);
      run<2, 1>(G, W, R, iters);
    }
  }
  return 0;
}
