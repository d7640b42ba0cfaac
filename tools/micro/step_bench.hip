// Stand-alone model of the forward IC(0) sweep step of ONE band (one wave alone on a CU), with the
// parts switchable, to attribute the per-step time.  Development aid.
//   F bits: 1 loads (3 streams, one 8-step block ahead)  2 result store  4 DPP shift
//           128 mask from the sign of pre (no byte stream)  256 stores batched at the block end  512 stores one block late
//           1024 with 512: loads first, then the late stores, scheduling barriers around the compute phase
//           8 LDS carry write  16 LDS boundary read per block  32 mask AND  64 two-block-ahead prefetch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define BLK 8
__device__ __forceinline__ double shr1(double v, double edge) {
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(edge), __double2loint(v), 0x138, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(edge), __double2hiint(v), 0x138, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
struct Ops { double in[BLK], pre[BLK]; int m[BLK]; };
template <int F>
__global__ __launch_bounds__(64) void k(const double* __restrict__ r, const double* __restrict__ pre, const int8_t* __restrict__ fm,
                                         double* __restrict__ q, int nblk, long long* cyc) {
  __shared__ double s_bnd[2][BLK];
  __shared__ double s_pub[BLK][64];
  const int lane = threadIdx.x;
  const double* p_in = r + lane; const double* p_pre = pre + lane; const int8_t* p_fm = fm + lane; double* p_out = q + lane + BLK * 64;
  Ops A, B;
  auto fetch = [&](Ops& o) {
#pragma unroll
    for (int j = 0; j < BLK; ++j) { o.in[j] = p_in[j * 64]; o.pre[j] = p_pre[j * 64]; o.m[j] = (F & 128) ? 0 : (int)p_fm[j * 64]; }
    p_in += BLK * 64; p_pre += BLK * 64; p_fm += BLK * 64;
  };
  fetch(A);
  if (!(F & 1)) fetch(B);
  double own = -0.0, out = -0.0, bnd0 = -0.0;
  s_bnd[0][lane & 7] = bnd0; s_bnd[1][lane & 7] = bnd0;
  double late[BLK];
#pragma unroll
  for (int j = 0; j < BLK; ++j) late[j] = 0.0;
  auto run = [&](auto par, Ops& cur, Ops& nxt) {
    constexpr int PAR = decltype(par)::value;
    if ((F & 512) && !(F & 1024)) {
#pragma unroll
      for (int j = 0; j < BLK; ++j) p_out[(j - BLK) * 64] = late[j];
    }
    if (F & 1) fetch(nxt);
    if ((F & 512) && (F & 1024)) {
#pragma unroll
      for (int j = 0; j < BLK; ++j) p_out[(j - BLK) * 64] = late[j];
    }
    if (F & 1024) __builtin_amdgcn_sched_barrier(0);
    double resv[BLK];
    double be[BLK];
#pragma unroll
    for (int j = 1; j < BLK; ++j) be[j] = (F & 16) ? s_bnd[PAR][j] : bnd0;
    be[0] = bnd0;
#pragma unroll
    for (int j = 0; j < BLK; ++j) {
      const double nbv = (F & 4) ? shr1(out, be[j]) : be[j] + out;
      const double t = cur.in[j] - own - nbv;
      const double cpre = (F & 128) ? fabs(cur.pre[j]) : cur.pre[j];
      const double qv = t * cpre;
      const int cm = (F & 128) ? (__double2hiint(cur.pre[j]) >> 31) : ((F & 32) ? cur.m[j] : -1);
      const double res = __hiloint2double(__double2hiint(qv) & cm, __double2loint(qv) & cm);
      const double carry = -1.0 * cpre * res;
      resv[j] = res;
      if ((F & 2) && !(F & 768)) p_out[j * 64] = res;
      own = carry; out = carry;
      if (F & 8) s_pub[j][lane] = carry;
    }
    if (F & 1024) __builtin_amdgcn_sched_barrier(0);
    if (F & 256) {
#pragma unroll
      for (int j = 0; j < BLK; ++j) p_out[j * 64] = resv[j];
    }
    if (F & 512) {
#pragma unroll
      for (int j = 0; j < BLK; ++j) late[j] = resv[j];
    }
    p_out += BLK * 64;
    if (F & 16) s_bnd[PAR ^ 1][lane & 7] = bnd0;
  };
  const long long t0 = clock64();
  for (int b = 0; b < nblk; b += 2) {
    run(std::integral_constant<int, 0>(), A, B);
    run(std::integral_constant<int, 1>(), B, A);
  }
  const long long t1 = clock64();
  if (!(F & 2)) q[lane] = own + out + s_pub[3][lane];
  if (lane == 0) cyc[0] = t1 - t0;
}
template <int F> void run(const double* r, const double* pre, const int8_t* fm, double* q, int nblk, long long* c) {
  long long h = 0;
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<F>, dim3(1), dim3(64), 0, 0, r, pre, fm, q, nblk, c);
  hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
  printf("F=%4d %s%s%s%s%s%s%s%s%s%s : %7.1f cycles/step  %6.1f ns/step\n", F, F & 1 ? "loads " : "", F & 2 ? "store " : "", F & 4 ? "dpp " : "",
         F & 8 ? "ldsW " : "", F & 16 ? "ldsR " : "", F & 32 ? "and " : "", F & 128 ? "signmask " : "", F & 256 ? "batchst " : "", F & 512 ? "latest " : "", F & 1024 ? "pinned " : "", (double)h / (nblk * BLK), (double)h / (nblk * BLK) / 2.4);
}
int main() {
  const int nblk = 512; const size_t n = (size_t)(nblk + 4) * BLK * 64;
  double *r, *pre, *q; int8_t* fm; long long* c;
  hipMalloc(&r, n * 8); hipMalloc(&pre, n * 8); hipMalloc(&q, n * 8); hipMalloc(&fm, n); hipMalloc(&c, 8);
  hipMemset(r, 0, n * 8); hipMemset(pre, 0, n * 8); hipMemset(fm, 0xff, n);
  hipMemset(pre, 0x80, n * 8);   // negative sign: "fluid" for the sign mask
  run<128 + 7>(r, pre, fm, q, nblk, c); run<128 + 31>(r, pre, fm, q, nblk, c);
  run<1024 + 128 + 7>(r, pre, fm, q, nblk, c); run<1024 + 128 + 31>(r, pre, fm, q, nblk, c);
  run<1024 + 128 + 256 + 7>(r, pre, fm, q, nblk, c); run<1024 + 128 + 256 + 31>(r, pre, fm, q, nblk, c);
  run<1024 + 128 + 512 + 3>(r, pre, fm, q, nblk, c); run<1024 + 128 + 512 + 7>(r, pre, fm, q, nblk, c); run<1024 + 128 + 512 + 31>(r, pre, fm, q, nblk, c);
  run<1024 + 128 + 5>(r, pre, fm, q, nblk, c); run<1024 + 128 + 29>(r, pre, fm, q, nblk, c);
  return 0;
}
