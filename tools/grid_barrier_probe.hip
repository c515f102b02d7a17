// grid_barrier_probe.hip -- what a grid barrier and its ingredients cost on MI355X (cooperative launch, 256-thread workgroups):
// latency of a returning agent-scope atomic, of an agent-scope load, of a write-through store + s_waitcnt; a flat barrier (all
// arrivals on one counter, spin on a flag word), a two-level one (8 groups by blockIdx % 8 -- the XCD round-robin -- each with its
// own counter and flag line), for several grid sizes.  Build: hipcc --offload-arch=gfx950 -O3 tools/grid_barrier_probe.hip -o /tmp/gbp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ uint32_t ld_dev(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_dev(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// mode 0: flat, spin on a flag; 1: flat, spin on the counter; 2: two-level (8 groups), flag per group; 3: two-level, all spin on ONE flag
__global__ __launch_bounds__(256) void barriers(uint32_t* sync, int reps, int mode, int sleep, unsigned long long* out) {
  const uint32_t G = gridDim.x;
  const uint32_t g = blockIdx.x & 7u, gsize = G / 8u;
  uint32_t epoch = 0;
  const unsigned long long t0 = wall_clock64();
  for (int r = 0; r < reps; ++r) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    ++epoch;
    if (threadIdx.x == 0) {
      if (mode == 0) {
        const uint32_t old = __hip_atomic_fetch_add(&sync[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1u == epoch * G) st_dev(&sync[32], epoch);
        else while (ld_dev(&sync[32]) < epoch) __builtin_amdgcn_s_sleep(2);
      } else if (mode == 1) {
        __hip_atomic_fetch_add(&sync[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (ld_dev(&sync[0]) < epoch * G) __builtin_amdgcn_s_sleep(2);
      } else {
        const uint32_t old = __hip_atomic_fetch_add(&sync[64 + 32 * g], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1u == epoch * gsize) {
          const uint32_t old2 = __hip_atomic_fetch_add(&sync[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (old2 + 1u == epoch * 8u) {
            if (mode == 2) { for (int k = 0; k < 8; ++k) st_dev(&sync[320 + 32 * k], epoch); }
            else st_dev(&sync[32], epoch);
          }
        }
        const uint32_t* flag = mode == 2 ? &sync[320 + 32 * g] : &sync[32];
        if (sleep == 0) while (ld_dev(flag) < epoch) { }
        else while (ld_dev(flag) < epoch) __builtin_amdgcn_s_sleep(2);
      }
    }
    __syncthreads();
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = wall_clock64() - t0;
}

__global__ void latencies(uint32_t* buf, unsigned long long* out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  uint32_t v = 0;
  unsigned long long t0 = wall_clock64();
  for (int k = 0; k < 200; ++k) v += __hip_atomic_fetch_add(&buf[1024], 1u + (v & 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  unsigned long long t1 = wall_clock64();
  out[1] = t1 - t0;
  uint32_t idx = 0;
  t0 = wall_clock64();
  for (int k = 0; k < 200; ++k) idx = ld_dev(&buf[2048 + (idx & 1023u)]);      // dependent agent-scope loads (the buffer holds zeros)
  t1 = wall_clock64();
  out[2] = t1 - t0 + idx;
  t0 = wall_clock64();
  for (int k = 0; k < 200; ++k) { st_dev(&buf[4096 + 32 * (k & 7)], (uint32_t)k); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
  t1 = wall_clock64();
  out[3] = t1 - t0;
  t0 = wall_clock64();
  for (int k = 0; k < 200; ++k) idx = buf[8192 + (idx & 1023u) + (k & 1)];        // dependent plain loads (L2 hits after the first)
  t1 = wall_clock64();
  out[4] = t1 - t0 + idx;
  out[5] = v;
}

int main() {
  uint32_t* sync; unsigned long long* out; uint32_t* buf;
  CHECK(hipMalloc(&sync, 4096 * 4)); CHECK(hipMalloc(&out, 64)); CHECK(hipMalloc(&buf, 1 << 20));
  CHECK(hipMemset(buf, 0, 1 << 20));
  unsigned long long h[8];
  latencies<<<1, 64>>>(buf, out);
  CHECK(hipDeviceSynchronize());
  CHECK(hipMemcpy(h, out, 64, hipMemcpyDeviceToHost));
  printf("returning agent-scope atomic: %.0f ns; agent-scope load: %.0f ns; write-through store + s_waitcnt: %.0f ns; plain load (L2): %.0f ns\n",
         h[1] * 10.0 / 200, h[2] * 10.0 / 200, h[3] * 10.0 / 200, h[4] * 10.0 / 200);
  const int reps = 500;
  for (int G : {64, 128, 256, 472, 512}) {
    for (int mode = 0; mode < 4; ++mode) {
      for (int sleep : {1, 0}) {
        if (sleep == 0 && mode < 2) continue;
        CHECK(hipMemset(sync, 0, 4096 * 4));
        int r = reps, m = mode, sl = sleep;
        void* args[] = {&sync, &r, &m, &sl, &out};
        CHECK(hipLaunchCooperativeKernel(reinterpret_cast<const void*>(barriers), dim3(G), dim3(256), args, 0, 0));
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(h, out, 8, hipMemcpyDeviceToHost));
        printf("G %3d mode %d sleep %d: %.2f us per barrier\n", G, mode, sleep, h[0] * 0.01 / reps);
      }
    }
  }
  return 0;
}
