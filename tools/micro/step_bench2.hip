// Second model of the forward step (one wave alone on a CU): how much does the ISSUE of the memory
// instructions cost?  Variants of where the operands come from / the results go.  Development aid.
//  M = 0  global loads/stores, 64-bit per-lane addresses (the kernel today)
//  M = 1  global loads/stores, SGPR base + 32-bit lane offset (saddr form)
//  M = 2  operands read from LDS (ds_read2_b64), results written to LDS only (helper waves would move them)
//  M = 3  like 2 but the result row is the carry row (backward: one ds_write per step in total)
//  M = 4  no memory at all
//  M = 5  global, records paired per lane: one 16-byte load per stream and TWO steps, one 16-byte store per two steps
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define BLK 8
__device__ __forceinline__ double shr1(double v, double edge) {
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(edge), __double2loint(v), 0x138, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(edge), __double2hiint(v), 0x138, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
typedef double d2g __attribute__((ext_vector_type(2)));
struct Ops { double in[BLK], pre[BLK]; d2g in2[4], pre2[4]; unsigned fb; };
template <int M>
__global__ __launch_bounds__(256) void k(const double* __restrict__ r, const double* __restrict__ pre, const unsigned* __restrict__ fbv,
                                         double* __restrict__ q, int nblk, long long* cyc) {
  __shared__ double s_in_all[4][2][BLK][2][64];
  __shared__ double s_out_all[4][BLK][64];
  __shared__ double s_pub_all[4][BLK][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  auto& s_in = s_in_all[wv]; auto& s_out = s_out_all[wv]; auto& s_pub = s_pub_all[wv];
  const size_t woff = (size_t)wv * (size_t)(nblk + 4) * BLK * 64;
  r += woff; pre += woff; q += woff; fbv += woff;
  const unsigned voff = lane * 8;
  const char* b_in = (const char*)r; const char* b_pre = (const char*)pre; char* b_out = (char*)q;
  const double* p_in = r + lane; const double* p_pre = pre + lane; double* p_out = q + lane; const unsigned* p_fb = fbv + lane;
  for (int j = 0; j < BLK; ++j) { s_in[0][j][0][lane] = 1.0; s_in[0][j][1][lane] = 0.5; s_in[1][j][0][lane] = 1.0; s_in[1][j][1][lane] = 0.5; }
  Ops A, B;
  auto fetch = [&](Ops& o, int slot) {
    if (M == 0) {
      asm volatile("global_load_dword %0, %1, off" : "=&v"(o.fb) : "v"(p_fb) : "memory");
#define LD(J) asm volatile("global_load_dwordx2 %0, %1, off offset:%2" : "=&v"(o.in[J]) : "v"(p_in), "n"((J) * 512)); \
              asm volatile("global_load_dwordx2 %0, %1, off offset:%2" : "=&v"(o.pre[J]) : "v"(p_pre), "n"((J) * 512));
      LD(0) LD(1) LD(2) LD(3) LD(4) LD(5) LD(6) LD(7)
#undef LD
      p_in += BLK * 64; p_pre += BLK * 64; p_fb += 64;
    } else if (M == 1) {
      asm volatile("global_load_dword %0, %1, off" : "=&v"(o.fb) : "v"(p_fb) : "memory");
#define LD(J) asm volatile("global_load_dwordx2 %0, %1, %2 offset:%3" : "=&v"(o.in[J]) : "v"(voff), "s"(b_in), "n"((J) * 512)); \
              asm volatile("global_load_dwordx2 %0, %1, %2 offset:%3" : "=&v"(o.pre[J]) : "v"(voff), "s"(b_pre), "n"((J) * 512));
      LD(0) LD(1) LD(2) LD(3) LD(4) LD(5) LD(6) LD(7)
#undef LD
      b_in += BLK * 512; b_pre += BLK * 512; p_fb += 64;
    } else if (M == 5) {
      asm volatile("global_load_dword %0, %1, off" : "=&v"(o.fb) : "v"(p_fb) : "memory");
      const char* q_in = b_in + lane * 16;      // pair layout: lane l owns 16 bytes per pair of steps
      const char* q_pre = b_pre + lane * 16;
#define LD2(P) { asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=&v"(o.in2[P]) : "v"(q_in), "n"((P) * 1024)); \
                 asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=&v"(o.pre2[P]) : "v"(q_pre), "n"((P) * 1024)); }
      LD2(0) LD2(1) LD2(2) LD2(3)
#undef LD2
      b_in += BLK * 512; b_pre += BLK * 512; p_fb += 64;
    } else if (M == 2 || M == 3) {
      o.fb = 0xff;
#pragma unroll
      for (int j = 0; j < BLK; ++j) { o.in[j] = s_in[slot][j][0][lane]; o.pre[j] = s_in[slot][j][1][lane]; }
    } else { o.fb = 0xff; for (int j = 0; j < BLK; ++j) { o.in[j] = 1.0; o.pre[j] = 0.5; } }
  };
  fetch(A, 0);
  double own = -0.0, out = -0.0;
  auto run = [&](Ops& cur, Ops& nxt, int slot) {
    fetch(nxt, slot);
    __builtin_amdgcn_sched_barrier(0);
    double prev_res = 0.0;
    auto step = [&](auto jc) {
      constexpr int j = decltype(jc)::value;
      if (M < 2) asm volatile("s_waitcnt vmcnt(%3)" : "+v"(cur.in[j]), "+v"(cur.pre[j]), "+v"(cur.fb) : "n"((7 - j) * 2 + 17 + j) : "memory");
      if (M == 5 && !(j & 1)) asm volatile("s_waitcnt vmcnt(%3)" : "+v"(cur.in2[j / 2]), "+v"(cur.pre2[j / 2]), "+v"(cur.fb) : "n"((3 - j / 2) * 2 + 9 + j / 2) : "memory");
      const double nbv = shr1(out, -0.0);
      const double cin = M == 5 ? ((j & 1) ? cur.in2[j / 2].y : cur.in2[j / 2].x) : cur.in[j];
      const double cpre = M == 5 ? ((j & 1) ? cur.pre2[j / 2].y : cur.pre2[j / 2].x) : cur.pre[j];
      const double t = cin - own - nbv;
      const double qv = t * cpre;
      const int cm = (int)(cur.fb << (31 - j)) >> 31;
      const double res = __hiloint2double(__double2hiint(qv) & cm, __double2loint(qv) & cm);
      const double carry = -1.0 * cpre * res;
      if (M == 0) p_out[j * 64] = res;
      if (M == 5) { if (j & 1) { d2g pr = {prev_res, res}; *reinterpret_cast<d2g*>(b_out + (j / 2) * 1024 + lane * 16) = pr; } prev_res = res; }
      if (M == 1) asm volatile("global_store_dwordx2 %0, %1, %2 offset:%3" :: "v"(voff), "v"(res), "s"(b_out), "n"(j * 512) : "memory");
      if (M == 2) s_out[j][lane] = res;
      own = carry; out = carry;
      if (M != 4) s_pub[j][lane] = carry;
    };
    step(std::integral_constant<int, 0>()); step(std::integral_constant<int, 1>()); step(std::integral_constant<int, 2>()); step(std::integral_constant<int, 3>());
    step(std::integral_constant<int, 4>()); step(std::integral_constant<int, 5>()); step(std::integral_constant<int, 6>()); step(std::integral_constant<int, 7>());
    __builtin_amdgcn_sched_barrier(0);
    p_out += BLK * 64; b_out += BLK * 512;
  };
  const long long t0 = clock64();
  for (int b = 0; b < nblk; b += 2) { run(A, B, 1); run(B, A, 0); }
  const long long t1 = clock64();
  q[lane] = own + out + s_pub[3][lane] + s_out[2][lane];
  if (lane == 0) cyc[wv] = t1 - t0;
}
template <int M> void run(const char* name, const double* r, const double* pre, const unsigned* fb, double* q, int nblk, long long* c) {
  for (int nw = 1; nw <= (M == 5 ? 1 : 4); ++nw) {   // (the paired-record variant is written for one wave)
    long long h[4] = {0, 0, 0, 0};
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<M>, dim3(1), dim3(64 * nw), 0, 0, r, pre, fb, q, nblk, c);
    hipMemcpy(h, c, 32, hipMemcpyDeviceToHost);
    printf("M=%d %-46s waves/CU %d: %7.1f cycles/step  %6.1f ns/step (wave 0)\n", M, name, nw, (double)h[0] / (nblk * BLK), (double)h[0] / (nblk * BLK) / 2.4);
  }
}
int main() {
  setvbuf(stdout, NULL, _IONBF, 0);
  const int nblk = 512; const size_t n = (size_t)(nblk + 4) * BLK * 64;
  double *r, *pre, *q; unsigned* fb; long long* c;
  hipMalloc(&r, 4 * n * 8); hipMalloc(&pre, 4 * n * 8); hipMalloc(&q, 4 * n * 8); hipMalloc(&fb, 4 * n * 4); hipMalloc(&c, 32);
  hipMemset(r, 0, 4 * n * 8); hipMemset(pre, 0, 4 * n * 8); hipMemset(fb, 0xff, 4 * n * 4);
  run<0>("global, 64-bit lane addresses", r, pre, fb, q, nblk, c);
  run<2>("operands and results through LDS", r, pre, fb, q, nblk, c);
  run<3>("operands through LDS, carry row = result row", r, pre, fb, q, nblk, c);
  run<4>("no memory", r, pre, fb, q, nblk, c);
  run<5>("global, paired records: 16-byte loads / stores", r, pre, fb, q, nblk, c);
  return 0;
}
