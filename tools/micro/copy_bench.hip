// copy_bench: which plain device-to-device copy reaches the HBM ceiling on this GPU (euler_measure_copy_bandwidth's probe)?
// build: hipcc --offload-arch=gfx950 -O3 -o copy_bench copy_bench.hip ; run: ./copy_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(256) void k_stride(const float4* __restrict__ s, float4* __restrict__ d, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
}
template <int U>
__global__ __launch_bounds__(256) void k_unroll(const float4* __restrict__ s, float4* __restrict__ d, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + (U - 1) * stride < n; i += U * stride) {
    float4 v[U];
#pragma unroll
    for (int k = 0; k < U; ++k) v[k] = s[i + k * stride];
#pragma unroll
    for (int k = 0; k < U; ++k) d[i + k * stride] = v[k];
  }
  for (; i < n; i += stride) d[i] = s[i];
}
typedef float f4v __attribute__((ext_vector_type(4)));
template <int U>
__global__ __launch_bounds__(256) void k_unroll_nt(const float4* __restrict__ s4, float4* __restrict__ d4, size_t n) {
  const f4v* s = reinterpret_cast<const f4v*>(s4); f4v* d = reinterpret_cast<f4v*>(d4);
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + (U - 1) * stride < n; i += U * stride) {
    f4v v[U];
#pragma unroll
    for (int k = 0; k < U; ++k) v[k] = __builtin_nontemporal_load(&s[i + k * stride]);
#pragma unroll
    for (int k = 0; k < U; ++k) __builtin_nontemporal_store(v[k], &d[i + k * stride]);
  }
  for (; i < n; i += stride) d[i] = s[i];
}
// one block = one contiguous chunk (each thread walks 16 B x 256 threads = 4 KB lines of its block's chunk)
template <int U>
__global__ __launch_bounds__(256) void k_chunk(const float4* __restrict__ s, float4* __restrict__ d, size_t n) {
  const size_t per = (n + gridDim.x - 1) / gridDim.x;
  const size_t lo = (size_t)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
  size_t i = lo + threadIdx.x;
  for (; i + (U - 1) * 256 < hi; i += U * 256) {
    float4 v[U];
#pragma unroll
    for (int k = 0; k < U; ++k) v[k] = s[i + k * 256];
#pragma unroll
    for (int k = 0; k < U; ++k) d[i + k * 256] = v[k];
  }
  for (; i < hi; i += 256) d[i] = s[i];
}
template <typename F> static double run(const char* name, F launch, size_t bytes) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  launch(); hipDeviceSynchronize();
  hipEventRecord(a); for (int r = 0; r < 10; ++r) launch(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double g = 2.0 * bytes * 10 / (ms * 1e-3) / 1e9;
  printf("%-28s %8.1f GB/s\n", name, g);
  return g;
}
int main() {
  const size_t bytes = (size_t)1 << 30, n = bytes / 16;
  float4 *s, *d; hipMalloc(&s, bytes); hipMalloc(&d, bytes); hipMemset(s, 1, bytes); hipMemset(d, 0, bytes);
  for (int grid : {1024, 2048, 4096, 8192, 16384, 65536}) {
    printf("grid %d\n", grid);
    run(" stride", [&] { hipLaunchKernelGGL(k_stride, dim3(grid), dim3(256), 0, 0, s, d, n); }, bytes);
    run(" unroll4", [&] { hipLaunchKernelGGL(k_unroll<4>, dim3(grid), dim3(256), 0, 0, s, d, n); }, bytes);
    run(" unroll8", [&] { hipLaunchKernelGGL(k_unroll<8>, dim3(grid), dim3(256), 0, 0, s, d, n); }, bytes);
    run(" unroll4 nt", [&] { hipLaunchKernelGGL(k_unroll_nt<4>, dim3(grid), dim3(256), 0, 0, s, d, n); }, bytes);
    run(" unroll8 nt", [&] { hipLaunchKernelGGL(k_unroll_nt<8>, dim3(grid), dim3(256), 0, 0, s, d, n); }, bytes);
    run(" chunk4", [&] { hipLaunchKernelGGL(k_chunk<4>, dim3(grid), dim3(256), 0, 0, s, d, n); }, bytes);
    run(" chunk8", [&] { hipLaunchKernelGGL(k_chunk<8>, dim3(grid), dim3(256), 0, 0, s, d, n); }, bytes);
  }
  run("hipMemcpyDtoD", [&] { hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, 0); }, bytes);
  return 0;
}
