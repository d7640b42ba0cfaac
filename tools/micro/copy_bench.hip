// copy_bench: which plain device-to-device copy reaches the HBM ceiling on this GPU (euler_measure_copy_bandwidth's probe)?
// build: hipcc --offload-arch=gfx950 -O3 -o copy_bench copy_bench.hip ; run: ./copy_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
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
// read-only / write-only streams (what does each direction reach alone?)
template <int U>
__global__ __launch_bounds__(256) void k_read(const float4* __restrict__ s4, float* __restrict__ sink, size_t n) {
  const f4v* s = reinterpret_cast<const f4v*>(s4);
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  f4v acc = {0.f, 0.f, 0.f, 0.f};
  for (; i + (U - 1) * stride < n; i += U * stride) {
    f4v v[U];
#pragma unroll
    for (int k = 0; k < U; ++k) v[k] = __builtin_nontemporal_load(&s[i + k * stride]);
#pragma unroll
    for (int k = 0; k < U; ++k) acc += v[k];
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.f) sink[0] = acc.x;
}
__global__ __launch_bounds__(256) void k_write(float4* __restrict__ d4, size_t n) {
  f4v* d = reinterpret_cast<f4v*>(d4);
  const f4v v = {1.f, 2.f, 3.f, 4.f};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) __builtin_nontemporal_store(v, &d[i]);
}
// each XCD (blockIdx % 8) streams its own contiguous eighth of the buffers
template <int U>
__global__ __launch_bounds__(256) void k_xcd(const float4* __restrict__ s4, float4* __restrict__ d4, size_t n) {
  const f4v* s = reinterpret_cast<const f4v*>(s4); f4v* d = reinterpret_cast<f4v*>(d4);
  const unsigned xcd = blockIdx.x & 7, bi = blockIdx.x >> 3, nb = gridDim.x >> 3;
  const size_t per = n / 8, lo = xcd * per, hi = lo + per;
  const size_t stride = (size_t)nb * 256;
  size_t i = lo + (size_t)bi * 256 + threadIdx.x;
  for (; i + (U - 1) * stride < hi; i += U * stride) {
    f4v v[U];
#pragma unroll
    for (int k = 0; k < U; ++k) v[k] = __builtin_nontemporal_load(&s[i + k * stride]);
#pragma unroll
    for (int k = 0; k < U; ++k) __builtin_nontemporal_store(v[k], &d[i + k * stride]);
  }
  for (; i < hi; i += stride) d[i] = s[i];
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
int main(int argc, char** argv) {
  const size_t mb = argc > 1 ? (size_t)atol(argv[1]) : 1024;
  const size_t bytes = mb << 20, n = bytes / 16;
  printf("buffer %zu MiB each\n", mb);
  float4 *s, *d; hipMalloc(&s, bytes); hipMalloc(&d, bytes); hipMemset(s, 1, bytes); hipMemset(d, 0, bytes);
  float* sink; hipMalloc(&sink, 64);
  for (int grid : {1024, 2048, 4096, 8192, 16384, 65536}) {
    printf("grid %d\n", grid);
    run(" stride", [&] { hipLaunchKernelGGL(k_stride, dim3(grid), dim3(256), 0, 0, s, d, n); }, bytes);
    run(" unroll4", [&] { hipLaunchKernelGGL(k_unroll<4>, dim3(grid), dim3(256), 0, 0, s, d, n); }, bytes);
    run(" unroll8", [&] { hipLaunchKernelGGL(k_unroll<8>, dim3(grid), dim3(256), 0, 0, s, d, n); }, bytes);
    run(" unroll4 nt", [&] { hipLaunchKernelGGL(k_unroll_nt<4>, dim3(grid), dim3(256), 0, 0, s, d, n); }, bytes);
    run(" unroll8 nt", [&] { hipLaunchKernelGGL(k_unroll_nt<8>, dim3(grid), dim3(256), 0, 0, s, d, n); }, bytes);
    run(" xcd8 nt", [&] { hipLaunchKernelGGL(k_xcd<8>, dim3(grid), dim3(256), 0, 0, s, d, n); }, bytes);
    printf("  (read / write alone count one direction: halve nothing, the figure printed is 2x bytes / time -> read the half)\n");
    run(" read8 nt (x0.5)", [&] { hipLaunchKernelGGL(k_read<8>, dim3(grid), dim3(256), 0, 0, s, sink, n); }, bytes);
    run(" write nt (x0.5)", [&] { hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, d, n); }, bytes);
    run(" chunk4", [&] { hipLaunchKernelGGL(k_chunk<4>, dim3(grid), dim3(256), 0, 0, s, d, n); }, bytes);
    run(" chunk8", [&] { hipLaunchKernelGGL(k_chunk<8>, dim3(grid), dim3(256), 0, 0, s, d, n); }, bytes);
  }
  run("hipMemcpyDtoD", [&] { hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, 0); }, bytes);
  return 0;
}
