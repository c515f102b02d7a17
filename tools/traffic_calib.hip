// traffic_calib.hip -- known-byte streaming kernels in the access shapes of the ICP iteration kernels, for calibrating
// rocprofv3's FETCH_SIZE / WRITE_SIZE on this GPU (MI355X_MICROARCH.md, HBM section: FETCH_SIZE reports half the bytes of a
// 16 B/lane streaming read; other widths and WRITE_SIZE are uncalibrated).  Not part of the product: built and run by
// tools/traffic_calib.sh on the GPU box.
//   calib_certify     per element: 12-byte row (global_load_dwordx3) + int + float read, one float written  -- nn_certify's shape
//   calib_accumulate  per element: 12-byte row + int + float read, nothing written                          -- accumulate's stream
//   calib_x4          per element: one 16-byte row read, one 16-byte row written                            -- the guide's own case
//   calib_scatter     per listed query (one in 50 elements, a hashed position): 12-byte row + int read, three dwords written to
//                     three arrays at that position                                                         -- nn_ball_listed's shape
//                     (known REQUESTED bytes; what the counters report per query is the granularity the memory system moves)
//   calib_certify_v4  calib_certify with four consecutive elements per lane (the same bytes in 16-byte accesses)
//   calib_certify_gather  calib_certify + two 16-byte gathers (matched point and normal) from the element's pair's own 21 700-row
//                     tables at a UNIFORMLY RANDOM row, workgroups striding over all pairs: what the gathers of nn_certify_acc would
//                     cost without the Morton order of the sources and the pair -> XCD mapping (no L2 locality at all)
// The durations of the counter-free pass (tools/traffic_calib.sh) are what each access shape alone costs on this GPU.
// Arrays are 256 x 120 000 elements (a 256-pair launch) and far larger than the 256 MiB Infinity Cache taken together.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_)); std::exit(1); } } while (0)

__global__ __launch_bounds__(256) void calib_certify(const float* __restrict__ src3, const int* __restrict__ idx, const float* __restrict__ lb,
                                                     float* __restrict__ d2, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float3 v = *reinterpret_cast<const float3*>(src3 + 3 * i);
    d2[i] = v.x + v.y + v.z + (float)idx[i] + lb[i];
  }
}
__global__ __launch_bounds__(256) void calib_certify_gather(const float* __restrict__ src3, const int* __restrict__ idx, const float* __restrict__ lb,
                                                            const float4* __restrict__ tq, const float4* __restrict__ tn, float* __restrict__ d2, size_t n) {
  const int kRows = 21700;                                 // the bench's CalculateNormals target
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float3 v = *reinterpret_cast<const float3*>(src3 + 3 * i);
    const size_t pair = i / 120000;
    unsigned z = (unsigned)i * 2654435761u; z ^= z >> 15;
    const size_t row = pair * kRows + (z + (unsigned)idx[i]) % kRows;
    const float4 q = tq[row], m = tn[row];
    d2[i] = v.x * q.x + v.y * q.y + v.z * q.z + m.x + m.y + m.z + lb[i];
  }
}
// calib_certify with four consecutive elements per lane: the same bytes in 16-byte accesses (3 + 1 + 1 loads, one store)
__global__ __launch_bounds__(256) void calib_certify_v4(const float* __restrict__ src3, const int* __restrict__ idx, const float* __restrict__ lb,
                                                        float* __restrict__ d2, size_t n) {
  const size_t n4 = n / 4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const float4 a = reinterpret_cast<const float4*>(src3)[3 * i], b = reinterpret_cast<const float4*>(src3)[3 * i + 1], c = reinterpret_cast<const float4*>(src3)[3 * i + 2];
    const int4 j = reinterpret_cast<const int4*>(idx)[i];
    const float4 l = reinterpret_cast<const float4*>(lb)[i];
    float4 o;
    o.x = a.x + a.y + a.z + (float)j.x + l.x; o.y = a.w + b.x + b.y + (float)j.y + l.y;
    o.z = b.z + b.w + c.x + (float)j.z + l.z; o.w = c.y + c.z + c.w + (float)j.w + l.w;
    reinterpret_cast<float4*>(d2)[i] = o;
  }
}
__global__ __launch_bounds__(256) void calib_accumulate(const float* __restrict__ src3, const int* __restrict__ idx, const float* __restrict__ d2in,
                                                        float* __restrict__ out, size_t n) {
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float3 v = *reinterpret_cast<const float3*>(src3 + 3 * i);
    acc += v.x + v.y + v.z + (float)idx[i] + d2in[i];
  }
  if (acc == 12345.678f) out[0] = acc;      // never true: keeps the loads alive without a store
}
__global__ __launch_bounds__(256) void calib_x4(const float4* __restrict__ in, float4* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float4 v = in[i];
    v.x += 1.f;
    out[i] = v;
  }
}

__global__ __launch_bounds__(256) void calib_scatter(const float* __restrict__ src3, int* __restrict__ idx, float* __restrict__ lb,
                                                     float* __restrict__ d2, size_t n, size_t nq) {
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < nq; k += (size_t)gridDim.x * blockDim.x) {
    unsigned long long z = k * 0x9E3779B97F4A7C15ull;                // splitmix-style hash: positions spread over the whole array
    z ^= z >> 31; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 29;
    const size_t i = (size_t)(z % n);
    const float3 v = *reinterpret_cast<const float3*>(src3 + 3 * i);
    const int j = idx[i];
    d2[i] = v.x + v.y; idx[i] = j + 1; lb[i] = v.z;
  }
}

int main() {
  const size_t n = (size_t)256 * 120000;
  float *src3, *lb, *d2, *out;
  int* idx;
  float4 *a4, *b4, *tq, *tn;
  CK(hipMalloc(&tq, (size_t)256 * 21700 * 16)); CK(hipMalloc(&tn, (size_t)256 * 21700 * 16));
  CK(hipMemset(tq, 0, (size_t)256 * 21700 * 16)); CK(hipMemset(tn, 0, (size_t)256 * 21700 * 16));
  CK(hipMalloc(&src3, n * 12)); CK(hipMalloc(&idx, n * 4)); CK(hipMalloc(&lb, n * 4)); CK(hipMalloc(&d2, n * 4)); CK(hipMalloc(&out, 256));
  CK(hipMalloc(&a4, n * 16)); CK(hipMalloc(&b4, n * 16));
  CK(hipMemset(src3, 0, n * 12)); CK(hipMemset(idx, 0, n * 4)); CK(hipMemset(lb, 0, n * 4)); CK(hipMemset(d2, 0, n * 4));
  CK(hipMemset(a4, 0, n * 16)); CK(hipMemset(b4, 0, n * 16));
  CK(hipDeviceSynchronize());
  const int blocks = 256 * 8 * 4;
  for (int rep = 0; rep < 4; ++rep) {
    // a different kernel's arrays are streamed in between, so no launch finds its own in the Infinity Cache
    hipLaunchKernelGGL(calib_x4, dim3(blocks), dim3(256), 0, 0, a4, b4, n);
    hipLaunchKernelGGL(calib_certify, dim3(blocks), dim3(256), 0, 0, src3, idx, lb, d2, n);
    hipLaunchKernelGGL(calib_x4, dim3(blocks), dim3(256), 0, 0, a4, b4, n);
    hipLaunchKernelGGL(calib_accumulate, dim3(blocks), dim3(256), 0, 0, src3, idx, lb, out, n);
    hipLaunchKernelGGL(calib_x4, dim3(blocks), dim3(256), 0, 0, a4, b4, n);
    hipLaunchKernelGGL(calib_scatter, dim3(blocks), dim3(256), 0, 0, src3, idx, lb, d2, n, n / 50);
    hipLaunchKernelGGL(calib_x4, dim3(blocks), dim3(256), 0, 0, a4, b4, n);
    hipLaunchKernelGGL(calib_certify_v4, dim3(blocks), dim3(256), 0, 0, src3, idx, lb, d2, n);
    hipLaunchKernelGGL(calib_x4, dim3(blocks), dim3(256), 0, 0, a4, b4, n);
    hipLaunchKernelGGL(calib_certify_gather, dim3(blocks), dim3(256), 0, 0, src3, idx, lb, tq, tn, d2, n);
  }
  CK(hipDeviceSynchronize());
  std::printf("{\"elements\": %zu, \"calib_certify\": {\"read_bytes\": %zu, \"written_bytes\": %zu}, \"calib_accumulate\": {\"read_bytes\": %zu, "
              "\"written_bytes\": 0}, \"calib_x4\": {\"read_bytes\": %zu, \"written_bytes\": %zu}, \"calib_scatter\": {\"queries\": %zu, \"read_bytes\": %zu, "
              "\"written_bytes\": %zu}, \"calib_certify_gather\": {\"read_bytes\": %zu, \"written_bytes\": %zu, \"gathered_bytes\": %zu}, \"calib_certify_v4\": {\"read_bytes\": %zu, \"written_bytes\": %zu}}\n", n, n * 20, n * 4, n * 20, n * 16, n * 16, n / 50, (n / 50) * 16, (n / 50) * 12,
              n * 20, n * 4, n * 32, n * 20, n * 4);
  return 0;
}
