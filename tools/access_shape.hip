// access_shape.hip -- what the ACCESSES of the fused certificate pass cost on this GPU without its arithmetic: per element a 12-byte
// row + a 4-byte shadow word streamed in (two rounds ahead), two 12-byte gathers from the 16-byte rows of the pair's two target
// tables at the index the shadow word holds (one round ahead), a 4-byte distance streamed out, one LDS histogram atomic; workgroups
// of 256 threads x 32 rounds, a pair's workgroups on one XCD, four workgroups per CU (the pass's occupancy, forced here by a
// dynamic LDS request) or eight.  The gathered indices are SYNTHETIC: the 64 elements of a wave take `clusters` runs of
// consecutive target rows, `row_stride` rows apart, around a base that moves through the pair's table with the element's position
// -- Morton-consecutive queries match a few runs of a few grid rows; clusters = 64 is one separate row per lane; `dup` lanes share a
// row (a scan has 5.5 source points per target point).  Not part of the product.
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/access_shape tools/access_shape.hip && /tmp/access_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_)); std::exit(1); } } while (0)

constexpr int kPairs = 512, kNs = 120000, kNt = 21760, kThreads = 256, kRounds = 32, kBins = 2048;

__global__ void make_indices(uint32_t* mb, int clusters, int row_stride, int dup) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)kPairs * kNs) return;
  const uint32_t e = (uint32_t)(i % kNs), lane = (uint32_t)(i & 63), wave = (uint32_t)(i >> 6);
  const uint32_t per = 64u / (uint32_t)clusters;                       // lanes per run
  const uint32_t span = (uint32_t)clusters * (uint32_t)row_stride + per;
  uint32_t h = wave * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
  const uint32_t base = (uint32_t)((double)e / kNs * (double)(kNt - span - 64)) + (h & 63u);
  mb[i] = base + (lane / per) * (uint32_t)row_stride + (lane % per) / (uint32_t)dup;       // dup lanes share a row (5.5 queries per target point)
}

template <bool GATHER>
__global__ __launch_bounds__(kThreads) void shape(const float* __restrict__ src3, const uint32_t* __restrict__ mb, const float4* __restrict__ tq,
                                                  const float4* __restrict__ tn, float* __restrict__ d2, uint32_t* __restrict__ hist) {
  extern __shared__ uint32_t s_dyn[];                                   // (occupancy limiter; the first kBins words are the histogram)
  constexpr int nblk = (kNs + kThreads * kRounds - 1) / (kThreads * kRounds);
  const int x = blockIdx.x & 7, q = blockIdx.x >> 3, blk = q % nblk, pair = (q / nblk) * 8 + x;
  if (pair >= kPairs) return;
  for (int k = threadIdx.x; k < kBins; k += kThreads) s_dyn[k] = 0;
  __syncthreads();
  const size_t so = (size_t)pair * kNs, to = (size_t)pair * kNt;
  const int base = blk * kThreads * kRounds;
  auto at = [&](int r) { return min(base + r * kThreads + (int)threadIdx.x, kNs - 1); };
  auto ld3 = [&](const float4* t, uint32_t j) { const float3 v = *reinterpret_cast<const float3*>(t + to + j); return v; };
  int ic = at(0);
  float3 s_1 = *reinterpret_cast<const float3*>(src3 + 3 * (so + ic));
  uint32_t m_1 = mb[so + ic];
  ic = at(1);
  float3 s_2 = *reinterpret_cast<const float3*>(src3 + 3 * (so + ic));
  uint32_t m_2 = mb[so + ic];
  float3 t_1 = make_float3(0, 0, 0), n_1 = make_float3(0, 0, 0);
  if (GATHER) { t_1 = ld3(tq, m_1); n_1 = ld3(tn, m_1); }
#pragma unroll 2
  for (int r = 0; r < kRounds; ++r) {
    const int i = base + r * kThreads + (int)threadIdx.x;
    const float3 s = s_1, t = t_1, n = n_1;
    const uint32_t m = m_1;
    s_1 = s_2; m_1 = m_2;
    if (GATHER && r + 1 < kRounds) { t_1 = ld3(tq, m_1); n_1 = ld3(tn, m_1); }
    if (r + 2 < kRounds) { ic = at(r + 2); s_2 = *reinterpret_cast<const float3*>(src3 + 3 * (so + ic)); m_2 = mb[so + ic]; }
    if (i < kNs) {
      const float v = s.x + s.y + s.z + t.x + t.y + t.z + n.x + n.y + n.z + (float)m;
      d2[so + i] = v;
      atomicAdd(&s_dyn[(__float_as_uint(v) >> 20) & (kBins - 1)], 1u);
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < kBins; k += kThreads) { const uint32_t v = s_dyn[k]; if (v) atomicAdd(&hist[(size_t)pair * kBins + k], v); }
}

__global__ __launch_bounds__(256) void sweep(const float4* __restrict__ in, float4* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}

int main() {
  const size_t n = (size_t)kPairs * kNs, nt = (size_t)kPairs * kNt;
  float *src3, *d2; uint32_t *mb, *hist; float4 *tq, *tn, *a4, *b4;
  CK(hipMalloc(&src3, n * 12)); CK(hipMalloc(&d2, n * 4)); CK(hipMalloc(&mb, n * 4)); CK(hipMalloc(&hist, (size_t)kPairs * kBins * 4));
  CK(hipMalloc(&tq, nt * 16)); CK(hipMalloc(&tn, nt * 16));
  const size_t nsw = (size_t)64 << 20;
  CK(hipMalloc(&a4, nsw * 16)); CK(hipMalloc(&b4, nsw * 16));
  CK(hipMemset(src3, 0, n * 12)); CK(hipMemset(tq, 0, nt * 16)); CK(hipMemset(tn, 0, nt * 16)); CK(hipMemset(hist, 0, (size_t)kPairs * kBins * 4)); CK(hipMemset(a4, 0, nsw * 16));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  constexpr int nblk = (kNs + kThreads * kRounds - 1) / (kThreads * kRounds);
  const int grid = nblk * 8 * (kPairs / 8);
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(shape<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 40 * 1024));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(shape<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 40 * 1024));
  std::printf("# tools/access_shape.hip on an MI355X: %d x %d elements, %d-row target tables, caches swept between launches; per element 12 + 4 B in, 4 B out,\n"
              "# two 12-byte gathers (synthetic indices: `clusters` runs of consecutive rows per wave, 150 rows apart), one LDS atomic; no arithmetic\n", kPairs, kNs, kNt);
  const int reps = 4;
  struct Case { int clusters; bool gather; int lds_kb; const char* what; int dup = 1; };
  const Case cases[] = {{1, false, 36, "no gathers, 4 workgroups / CU"}, {1, true, 36, "1 run per wave"}, {4, true, 36, "4 runs per wave"}, {8, true, 36, "8 runs per wave"},
                        {16, true, 36, "16 runs per wave"}, {64, true, 36, "64 separate rows per wave"}, {8, true, 8, "8 runs per wave, 8 workgroups / CU"},
                        {64, true, 8, "64 separate rows, 8 workgroups / CU"}, {1, false, 8, "no gathers, 8 workgroups / CU"},
                        {8, true, 36, "8 runs per wave, 4 lanes per row", 4}, {4, true, 36, "4 runs per wave, 4 lanes per row", 4},
                        {2, true, 36, "2 runs per wave, 4 lanes per row", 4}, {1, true, 36, "1 run per wave, 4 lanes per row", 4}};
  for (const Case& c : cases) {
    hipLaunchKernelGGL(make_indices, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, mb, c.clusters, 150, c.dup);
    double t = 0;
    for (int rep = 0; rep < reps + 1; ++rep) {
      hipLaunchKernelGGL(sweep, dim3(8192), dim3(256), 0, 0, a4, b4, nsw);
      CK(hipEventRecord(e0));
      if (c.gather) hipLaunchKernelGGL(shape<true>, dim3(grid), dim3(kThreads), (size_t)c.lds_kb * 1024, 0, src3, mb, tq, tn, d2, hist);
      else hipLaunchKernelGGL(shape<false>, dim3(grid), dim3(kThreads), (size_t)c.lds_kb * 1024, 0, src3, mb, tq, tn, d2, hist);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep > 0) t += ms;
    }
    CK(hipGetLastError());
    std::printf("%-40s %.3f ms per %d pairs = %.2f TB/s of the pass's 36.8 algorithmic B per element\n", c.what, t / reps, kPairs, n * 36.8 / (t / reps * 1e-3) / 1e12);
  }
  return 0;
}
