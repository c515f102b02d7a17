// stream_shapes.hip -- what the access SHAPE of the fused certificate pass's streams costs on this GPU (no arithmetic, no gathers):
// the source as 12-byte rows (global_load_dwordx3, the layout the kernels stream now) against three separate float arrays (SoA:
// three fully coalesced dword loads), each with the 4-byte shadow read and the 4-byte distance written.  Not part of the product.
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/stream_shapes tools/stream_shapes.hip && /tmp/stream_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_)); std::exit(1); } } while (0)

__global__ __launch_bounds__(256) void shape_rows12(const float* __restrict__ src3, const unsigned* __restrict__ mb, float* __restrict__ d2, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float3 v = *reinterpret_cast<const float3*>(src3 + 3 * i);
    d2[i] = v.x + v.y + v.z + (float)mb[i];
  }
}
__global__ __launch_bounds__(256) void shape_soa(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z,
                                                 const unsigned* __restrict__ mb, float* __restrict__ d2, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    d2[i] = x[i] + y[i] + z[i] + (float)mb[i];
}
// four consecutive elements per lane, SoA: every access 16 bytes
__global__ __launch_bounds__(256) void shape_soa_v4(const float4* __restrict__ x, const float4* __restrict__ y, const float4* __restrict__ z,
                                                    const uint4* __restrict__ mb, float4* __restrict__ d2, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const float4 a = x[i], b = y[i], c = z[i];
    const uint4 m = mb[i];
    d2[i] = make_float4(a.x + b.x + c.x + (float)m.x, a.y + b.y + c.y + (float)m.y, a.z + b.z + c.z + (float)m.z, a.w + b.w + c.w + (float)m.w);
  }
}
__global__ __launch_bounds__(256) void sweep(const float4* __restrict__ in, float4* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}

int main() {
  const size_t n = (size_t)512 * 120000;
  float *src3, *x, *y, *z, *d2; unsigned* mb; float4 *a4, *b4;
  CK(hipMalloc(&src3, n * 12)); CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&y, n * 4)); CK(hipMalloc(&z, n * 4)); CK(hipMalloc(&d2, n * 4)); CK(hipMalloc(&mb, n * 4));
  const size_t ns = (size_t)64 << 20;
  CK(hipMalloc(&a4, ns * 16)); CK(hipMalloc(&b4, ns * 16));
  CK(hipMemset(src3, 0, n * 12)); CK(hipMemset(x, 0, n * 4)); CK(hipMemset(y, 0, n * 4)); CK(hipMemset(z, 0, n * 4)); CK(hipMemset(mb, 0, n * 4)); CK(hipMemset(a4, 0, ns * 16));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int blocks = 256 * 8 * 4;
  double t[3] = {0, 0, 0};
  const int reps = 5;
  for (int rep = 0; rep < reps + 1; ++rep) {
    for (int k = 0; k < 3; ++k) {
      hipLaunchKernelGGL(sweep, dim3(blocks), dim3(256), 0, 0, a4, b4, ns);          // 2 GiB through the caches
      CK(hipEventRecord(e0));
      if (k == 0) hipLaunchKernelGGL(shape_rows12, dim3(blocks), dim3(256), 0, 0, src3, mb, d2, n);
      if (k == 1) hipLaunchKernelGGL(shape_soa, dim3(blocks), dim3(256), 0, 0, x, y, z, mb, d2, n);
      if (k == 2) hipLaunchKernelGGL(shape_soa_v4, dim3(blocks), dim3(256), 0, 0, (const float4*)x, (const float4*)y, (const float4*)z, (const uint4*)mb, (float4*)d2, n / 4);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep > 0) t[k] += ms;
    }
  }
  const char* names[3] = {"rows of 12 bytes + shadow -> d2", "SoA x, y, z + shadow -> d2", "SoA, four elements per lane"};
  for (int k = 0; k < 3; ++k) std::printf("%-36s %.3f ms per %zu elements = %.2f TB/s of 20 B per element\n", names[k], t[k] / reps, n, n * 20.0 / (t[k] / reps * 1e-3) / 1e12);
  return 0;
}
