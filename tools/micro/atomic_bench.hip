// Development aid: what sets the cost of the binning kernel's memory-side atomics - the number of lane operations or the
// number of distinct 64-byte lines per wave instruction?  16 active lanes per wave add 4 to (a) 16 cells of one COLUMN of a
// row-major grid (stride X: 16 lines), (b) 16 consecutive cells (one line).
#include <hip/hip_runtime.h>
#include <cstdio>
#define X 8192
#define Y 8192
__global__ void k_atomic(unsigned int* cnt, int transposed, int every) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // "marker" index, 4 per cell, cells walk a column
  const size_t cell = i >> 2;
  const size_t x = cell / (Y / 2), y = cell % (Y / 2);
  if (x >= X) return;
  const size_t c = transposed ? x * Y + y : y * X + x;
  if ((threadIdx.x & (every - 1)) == 0) atomicAdd(&cnt[c], (unsigned int)every);
}
int main() {
  unsigned int* cnt;
  hipMalloc(&cnt, (size_t)X * Y * 4);
  const size_t n = (size_t)X * (Y / 2) * 4;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int every = 4; every >= 1; every >>= 2)
    for (int tr = 0; tr < 2; ++tr) {
      hipMemset(cnt, 0, (size_t)X * Y * 4);
      hipLaunchKernelGGL(k_atomic, dim3(n / 256), dim3(256), 0, 0, cnt, tr, every);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k_atomic, dim3(n / 256), dim3(256), 0, 0, cnt, tr, every);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("lanes adding: every %d-th, %s layout: %.3f ms per pass (%zu lane atomics)\n", every, tr ? "column-major (transposed)" : "row-major", ms / 5, n / every);
    }
  return 0;
}
