// Dependent-chain latencies of the instructions on the IC(0) sweep's critical path (one wave alone
// on its SIMD).  Development aid: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off chain_latency.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 4096
__device__ __forceinline__ double shr1(double v, double edge) {
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(edge), __double2loint(v), 0x138, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(edge), __double2hiint(v), 0x138, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ int dpp32(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, CTRL, 0xf, 0xf, false); }
// lane l <- lane l-1 built from row-local moves: row_bcast:15 seeds lane 0 of rows 1..3, row_shr:1 does the rest
__device__ __forceinline__ double shr1_rows(double v, double edge) {
  int lo = dpp32<0x142>(__double2loint(edge), __double2loint(v));   // row_bcast:15
  int hi = dpp32<0x142>(__double2hiint(edge), __double2hiint(v));
  lo = dpp32<0x111>(lo, __double2loint(v));                          // row_shr:1
  hi = dpp32<0x111>(hi, __double2hiint(v));
  return __hiloint2double(hi, lo);
}
template <int MODE>
__global__ void k(double* out, long long* cyc, double a, double b, int m) {
  double x = out[threadIdx.x], y = x;
  const long long t0 = clock64();
  const unsigned long long w0 = wall_clock64();
#pragma unroll 1
  for (int i = 0; i < N / 16; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (MODE == 0) x = x * a;                                     // v_mul_f64 chain
      if (MODE == 1) x = x - a;                                     // v_add_f64 chain
      if (MODE == 2) x = shr1(x, a);                                // DPP pair chain
      if (MODE == 3) x = __hiloint2double(__double2hiint(x) & m, __double2loint(x) & m);   // v_and pair chain
      if (MODE == 4) {                                              // the forward sweep's step without memory
        const double nbv = shr1(y, a);
        const double t = b - y - nbv;
        const double q = t * a;
        const double r = __hiloint2double(__double2hiint(q) & m, __double2loint(q) & m);
        y = -1.0 * a * r;
        x += r;   // keep r alive off the chain
      }
      if (MODE == 5) { x = x * a; x = shr1(x, a); }                 // mul -> dpp
      if (MODE == 6) { x = x * a; x = x - b; }                      // mul -> add
      if (MODE == 7) x = fma(x, a, b);                              // v_fma_f64 chain
      if (MODE == 10) x = __hiloint2double(dpp32<0x111>(0, __double2hiint(x)), dpp32<0x111>(0, __double2loint(x)));   // row_shr:1 pair
      if (MODE == 11) x = shr1_rows(x, a);
      if (MODE == 12) x = __hiloint2double(__double2hiint(x), dpp32<0x138>(0, __double2loint(x)));   // one wave_shr:1
      if (MODE == 13) x = __hiloint2double(__double2hiint(x), dpp32<0x111>(0, __double2loint(x)));   // one row_shr:1
      if (MODE == 14) x = __hiloint2double(__double2hiint(x), dpp32<0x142>(0, __double2loint(x)));   // one row_bcast:15
      if (MODE == 15) {                                             // forward step with the row-built shift
        const double nbv = shr1_rows(y, a);
        const double t = b - y - nbv;
        const double q = t * a;
        const double r = __hiloint2double(__double2hiint(q) & m, __double2loint(q) & m);
        y = -1.0 * a * r;
        x += r;
      }
      if (MODE == 16) { x = x * a; x = __hiloint2double(__double2hiint(x), dpp32<0x111>(0, __double2loint(x))); }   // mul -> row_shr
      if (MODE == 8) { float f = (float)x; f = f * (float)a; x = f; } // cvt chain (reference point)
    }
  }
  const long long t1 = clock64();
  const unsigned long long w1 = wall_clock64();
  out[threadIdx.x] = x + y;
  if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = (long long)(w1 - w0); }
}
template <int MODE> void run(const char* name, int ops) {
  double* d; long long* c; long long h[2];
  hipMalloc(&d, 64 * 8); hipMalloc(&c, 16); hipMemset(d, 0, 64 * 8);
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64), 0, 0, d, c, 1.0000001, 0.5, -1);
  hipMemcpy(h, c, 16, hipMemcpyDeviceToHost);
  printf("%-28s %7.2f clk64/iter  %7.2f ns/iter  (%d dependent ops per iter)\n", name, (double)h[0] / N, (double)h[1] * 10.0 / N, ops);
  hipFree(d); hipFree(c);
}
int main() {
  run<0>("v_mul_f64", 1); run<1>("v_add_f64", 1); run<2>("dpp wave_shr pair", 1); run<3>("v_and pair", 1);
  run<4>("forward step (no memory)", 5); run<5>("mul->dpp", 2); run<6>("mul->add", 2); run<7>("v_fma_f64", 1); run<8>("cvt-mul-cvt f32", 3);
  run<10>("row_shr:1 pair", 1); run<11>("bcast15+row_shr pair", 2); run<12>("one wave_shr:1", 1); run<13>("one row_shr:1", 1); run<14>("one row_bcast:15", 1);
  run<15>("forward step, row-built shift", 6); run<16>("mul->row_shr", 2);
  return 0;
}
