// flag_latency_probe.hip — what does a point-to-point hand-off between two workgroups cost on MI355X? (r03, for the next round)
// The arrival-ticket fix-up and the grid-barrier probe showed that device-scope fences (L2 write-back) and contended counters are
// slower than a kernel boundary.  This probe measures the cheapest form: workgroup A produces `n` floats and raises a flag,
// workgroup B polls the flag, consumes the floats and raises its own flag; 2 000 round trips, several payload sizes,
//   method 0: plain stores + __threadfence() (release) / __threadfence() after the poll (acquire)
//   method 1: agent-scope relaxed atomic stores and loads for payload and flag, no fence (s_waitcnt before the flag)
//   method 2: 16-byte stores / loads with the sc0 sc1 bits (write-through / L2 bypass) through inline assembly, no fence
// for a pair on the same XCD (workgroups 0 and 8 of a round-robin dispatch) and on different XCDs (0 and 1).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP %s @%d\n", hipGetErrorString(e_), __LINE__); exit(2);} } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_wt(float* p, f32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ f32x4 load_bypass(const float* p) {
  f32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

template <int METHOD>
__global__ __launch_bounds__(256) void pingpong(unsigned* flags, float* buf, int n, int iters, int partner, unsigned long long* out) {
  const int me = blockIdx.x;
  if (me != 0 && me != partner) return;
  const int role = me == 0 ? 0 : 1;
  float* mine = buf + role * 65536;
  float* theirs = buf + (1 - role) * 65536;
  unsigned* my_flag = flags + role * 64;
  unsigned* their_flag = flags + (1 - role) * 64;
  float acc = 0.f;
  const unsigned long long t0 = wall_clock64();
  for (int i = 1; i <= iters; ++i) {
    if (role == 1) {  // B waits for A's i-th message first
      if (threadIdx.x == 0) while (__hip_atomic_load(their_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)i) __builtin_amdgcn_s_sleep(1);
      __syncthreads();
      if (METHOD == 0) __threadfence();
      if (METHOD == 2) {
        for (int j = 4 * threadIdx.x; j < n; j += 1024) { const f32x4 v = load_bypass(theirs + j); acc += v[0] + v[1] + v[2] + v[3]; }
      } else
      for (int j = threadIdx.x; j < n; j += 256) acc += METHOD == 0 ? theirs[j] : __hip_atomic_load(theirs + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (METHOD == 2) {
      for (int j = 4 * threadIdx.x; j < n; j += 1024) {
        const float b = acc + (float)(i + j);
        store_wt(mine + j, (f32x4){b, b + 1.f, b + 2.f, b + 3.f});
      }
    } else
    for (int j = threadIdx.x; j < n; j += 256) {
      if (METHOD == 0) mine[j] = acc + (float)(i + j);
      else __hip_atomic_store(mine + j, acc + (float)(i + j), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (METHOD == 0) __threadfence(); else __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(my_flag, (unsigned)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (role == 0) {  // A waits for B's answer
      if (threadIdx.x == 0) while (__hip_atomic_load(their_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)i) __builtin_amdgcn_s_sleep(1);
      __syncthreads();
      if (METHOD == 0) __threadfence();
      if (METHOD == 2) {
        for (int j = 4 * threadIdx.x; j < n; j += 1024) { const f32x4 v = load_bypass(theirs + j); acc += v[0] + v[1] + v[2] + v[3]; }
      } else
      for (int j = threadIdx.x; j < n; j += 256) acc += METHOD == 0 ? theirs[j] : __hip_atomic_load(theirs + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (threadIdx.x == 0 && role == 0) { out[0] = wall_clock64() - t0; out[1] = (unsigned long long)acc; }
}

int main() {
  unsigned* flags; float* buf; unsigned long long* out;
  CK(hipMalloc(&flags, 4096)); CK(hipMalloc(&buf, 2 * 65536 * 4)); CK(hipMalloc(&out, 64));
  const int iters = 2000;
  for (int partner : {8, 1}) for (int method : {0, 1, 2}) for (int n : {0, 256, 4096, 16384}) {
    CK(hipMemset(flags, 0, 4096)); CK(hipMemset(buf, 0, 2 * 65536 * 4));
    CK(hipDeviceSynchronize());
    if (method == 0) hipLaunchKernelGGL(pingpong<0>, dim3(16), dim3(256), 0, 0, flags, buf, n, iters, partner, out);
    else if (method == 1) hipLaunchKernelGGL(pingpong<1>, dim3(16), dim3(256), 0, 0, flags, buf, n, iters, partner, out);
    else hipLaunchKernelGGL(pingpong<2>, dim3(16), dim3(256), 0, 0, flags, buf, n, iters, partner, out);
    CK(hipDeviceSynchronize());
    unsigned long long h[2]; CK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost));
    // wall_clock64 ticks at 100 MHz
    printf("%s, %s, payload %5d floats: %.2f us per one-way hand-off\n", partner == 8 ? "same XCD (wg 0 <-> 8)" : "other XCD (wg 0 <-> 1)",
           method == 0 ? "plain + __threadfence" : method == 1 ? "agent-scope atomics  " : "sc0 sc1 dwordx4      ", n, (double)h[0] * 10.0 / 1000.0 / (2.0 * iters));
  }
  return 0;
}
