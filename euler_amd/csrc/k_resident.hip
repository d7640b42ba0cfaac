// k_resident.hip — the tile-local PCG solve (EULER_PRECOND_IC0_TILE; reference project(), main.c:709-767) as ONE persistent launch whose
// vectors never leave the register file.  For the grids whose every 16-record chunk finds a wave of its own on the chip at once
// (<= 2048 chunks = 2 M cells: BASELINE configs[1], the 1024^2 dam break, has 1152) the multi-kernel form is bound by latency, not by
// bytes: 1024^2 ran 31.6 us per iteration for 25 MB of algorithmic traffic (0.10 of the roofline), two launches whose waves each wait
// for a scalar, for their operands, for the reduction's ticket and for the next launch.  Here:
//
//   * one WAVE per active chunk (the solve's chunk list, k_build_system), 4 waves per workgroup, every workgroup resident at once (two per CU in double, three in float);
//     r, s, p and E^-1 of the chunk's 16 x 64 cells live in the wave's registers from the first iteration to the last (64 values per lane);
//   * A s needs the chunk's halo - the record before and after it (64 cells each), the row below lane 0 and above lane 63 (16 each): every
//     wave publishes those cells of z and s (write-through) before the iteration's second reduction; behind it the neighbours form the halo's
//     s' = z + beta s themselves with the owner's expression (the same bits - how the row-slab mode keeps its ghost rows, k_pcg.hip SLAB 2);
//   * the two reductions of an iteration (dot(s, A s) -> alpha; max |r| and dot(z, r) -> done, beta) are all-gathers of 16-byte
//     {value, generation} granules, one per workgroup, written by ONE write-through store and swept by every workgroup's first wave until
//     all carry the iteration's generation, then folded in workgroup order (the same bits everywhere, so alpha, beta and `done` need no broadcast).
//     A granule seen with its generation also vouches for the halo cells its workgroup published before it (each thread drains its stores first).
//
// Two grid-wide synchronisations and ~2 KB of halo per wave and iteration; no HBM traffic inside the solve at all.
// T = double: the multi-kernel tile mode's arithmetic, expression for expression; the sums fold per workgroup instead of per block, so the iterates
// agree to rounding (EULER_DOT_TREE is a tolerance mode everywhere).  T = float: BASELINE configs[1]'s "fp32" - every solver vector in float, sums
// and scalars in double (euler_config.pcg_precision; restated in the oracle: eo_sim.pcg_f32).  Not the reference's iterates (its PCG is double,
// main.c:577-578,716): a labelled variant.
//
// Safety: the waits are bounded (2 s); a workgroup that gives up raises *err and every workgroup leaves; the host then solves the same system
// with the multi-kernel path (b is untouched) and stops using this kernel on the handle.
#include "euler_dev.h"

#include <stdlib.h>
#include <type_traits>

#define RS_THREADS 256
#define RS_WAVES 4
#define RS_MAX_WG 768          // granules per kind; up to 3 workgroups per CU
#define RS_SHL1 0x130          // DPP wave shifts (k_pcg.hip): lane l <- lane l + 1, lane 63 <- the injected value
#define RS_SHR1 0x138          // lane l <- lane l - 1, lane 0 <- the injected value
#define RS_TIMEOUT_TICKS 200000000ull   // of the 100 MHz wall clock

struct ResArgs {
  SkewGeom g;
  const uint8_t* mask;
  const double* b;
  double *p, *r, *pre;
  void *zx, *sx;               // the cells other waves read: z and s at their band-skewed element index, as T (storage: the handle's z and s arrays)
  const unsigned int* list;
  PcgScalars* sc;
  unsigned long long* gran;    // [2 generations][2 groups][2 RS_MAX_WG] granules of 16 bytes (driver.hip allocates them)
  unsigned long long tag0;     // generation of this launch's first reduction (monotonic over the handle's life: nothing is ever cleared)
  int band_lo, max_iters;
  double tol;
  int* err;
  int force_fail;              // test hook (EULER_OPT_RESIDENT_FORCE_TIMEOUT): give up at once with error word 1, as if a wait had run out
};

namespace eu_resident {
typedef unsigned int rs_u4 __attribute__((ext_vector_type(4)));

template <int CTRL> __device__ __forceinline__ double rs_shift(double v, double edge) {
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(edge), __double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(edge), __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
template <int CTRL> __device__ __forceinline__ float rs_shift(float v, float edge) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(edge), __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ double rs_bcast(double v, int src) { return __shfl(v, src, 64); }
__device__ __forceinline__ float rs_bcast(float v, int src) { return __shfl(v, src, 64); }

// write-through stores / loads that bypass the non-coherent caches (MI355X_MICROARCH "valid forms": sc0 sc1 on both sides)
__device__ __forceinline__ void rs_st(double* p, double v) { asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void rs_st(float* p, float v) { asm volatile("global_store_dword %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void rs_ld6(const double* p0, const double* p1, const double* p2, const double* p3, const double* p4, const double* p5,
                                       double& a0, double& a1, double& a2, double& a3, double& a4, double& a5) {
  asm volatile("global_load_dwordx2 %0, %6, off sc0 sc1\n\tglobal_load_dwordx2 %1, %7, off sc0 sc1\n\tglobal_load_dwordx2 %2, %8, off sc0 sc1\n\t"
               "global_load_dwordx2 %3, %9, off sc0 sc1\n\tglobal_load_dwordx2 %4, %10, off sc0 sc1\n\tglobal_load_dwordx2 %5, %11, off sc0 sc1\n\t"
               "s_waitcnt vmcnt(0)"
               : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5) : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(p4), "v"(p5) : "memory");
}
__device__ __forceinline__ void rs_ld6(const float* p0, const float* p1, const float* p2, const float* p3, const float* p4, const float* p5,
                                       float& a0, float& a1, float& a2, float& a3, float& a4, float& a5) {
  asm volatile("global_load_dword %0, %6, off sc0 sc1\n\tglobal_load_dword %1, %7, off sc0 sc1\n\tglobal_load_dword %2, %8, off sc0 sc1\n\t"
               "global_load_dword %3, %9, off sc0 sc1\n\tglobal_load_dword %4, %10, off sc0 sc1\n\tglobal_load_dword %5, %11, off sc0 sc1\n\t"
               "s_waitcnt vmcnt(0)"
               : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5) : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(p4), "v"(p5) : "memory");
}
__device__ __forceinline__ void rs_st_gran(unsigned long long* g, double v, unsigned long long tag) {
  rs_u4 w;
  w.x = (unsigned int)__double2loint(v); w.y = (unsigned int)__double2hiint(v); w.z = (unsigned int)tag; w.w = (unsigned int)(tag >> 32);
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(g), "v"(w) : "memory");
}
__device__ __forceinline__ void rs_ld_gran4(const unsigned long long* p0, const unsigned long long* p1, const unsigned long long* p2, const unsigned long long* p3,
                                            rs_u4& a0, rs_u4& a1, rs_u4& a2, rs_u4& a3) {
  asm volatile("global_load_dwordx4 %0, %4, off sc0 sc1\n\tglobal_load_dwordx4 %1, %5, off sc0 sc1\n\tglobal_load_dwordx4 %2, %6, off sc0 sc1\n\t"
               "global_load_dwordx4 %3, %7, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
               : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3) : "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory");
}
__device__ __forceinline__ double rs_gran_val(const rs_u4& w) { return __hiloint2double((int)w.y, (int)w.x); }
__device__ __forceinline__ unsigned long long rs_gran_tag(const rs_u4& w) { return ((unsigned long long)w.w << 32) | w.z; }

// one cell of the E^-1 recurrence (main.c:586-600; k_pcg.hip factor_step), in T
template <typename T> __device__ __forceinline__ T rs_factor_step(T aa, T own, T nbv) {
  const T cl = (T)-1 * own, cb = (T)-1 * nbv;
  T e = aa - cl * cl - cb * cb;
  if (e < (T)0.25 * aa) e = (aa != (T)0) ? aa : (T)1;
  return (T)1 / (std::is_same<T, float>::value ? (T)sqrtf((float)e) : (T)sqrt((double)e));
}

// Grid-wide reductions of NV values (value n: a maximum if is_max[n], else a sum) in ONE sweep: every workgroup's partials go out as NV adjacent granules (index
// wg * NV + n), every workgroup's first wave gathers all nwg * NV of them - lane l takes the granules l, l + 64, ...: with NV = 2 the even lanes see only value 0, the
// odd lanes only value 1, so both values cost one round trip - and folds them in a fixed order (a lane its own in index order, then a butterfly over the lanes of
// the same parity).  Returns false on a timeout.  `group` 0 / 1 = the iteration's first / second reduction (separate granule arrays, two generations each).
template <int NV>
__device__ __forceinline__ bool rs_reduce(const ResArgs& a, int nwg, double (&v)[NV], const bool (&is_max)[NV], int group, unsigned long long tag, double (*s_red)[3], double* s_tot, int* s_fail) {
  static_assert(NV == 1 || NV == 2, "one or two values");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int n = 0; n < NV; ++n) {
    const double w = is_max[n] ? eu_wave_max(v[n]) : eu_wave_sum(v[n]);
    if (lane == 0) s_red[wave][n] = w;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this thread's halo cells have landed before the granule that vouches for them
  __syncthreads();
  unsigned long long* gbase = a.gran + (size_t)((int)(tag & 1) * 2 + group) * (RS_MAX_WG * 2) * 2;      // [generation parity][group][RS_MAX_WG * 2 granules] x 2 words
  if (threadIdx.x < NV) {
    const int n = threadIdx.x;
    double t = s_red[0][n];
    for (int k = 1; k < RS_WAVES; ++k) t = is_max[n] ? (s_red[k][n] > t ? s_red[k][n] : t) : t + s_red[k][n];
    rs_st_gran(gbase + ((size_t)blockIdx.x * NV + n) * 2, t, tag);
  }
  if (wave == 0) {
    const unsigned long long t_start = wall_clock64();
    const int total = nwg * NV;
    const bool mine_max = is_max[NV == 2 ? (lane & 1) : 0];      // (the kind of every granule this lane sees)
    bool failed = false;
    double acc = 0.0;
    for (int q = 0; 256 * q < total && !failed; ++q) {      // 4 granules per lane and pass
      const int i0 = lane + 256 * q;
      rs_u4 w0, w1, w2, w3;
      const bool h0 = i0 < total, h1 = i0 + 64 < total, h2 = i0 + 128 < total, h3 = i0 + 192 < total;
      const unsigned long long *p0 = gbase + (size_t)(h0 ? i0 : 0) * 2, *p1 = gbase + (size_t)(h1 ? i0 + 64 : 0) * 2,
                               *p2 = gbase + (size_t)(h2 ? i0 + 128 : 0) * 2, *p3 = gbase + (size_t)(h3 ? i0 + 192 : 0) * 2;
      for (unsigned int spins = 1;; ++spins) {
        rs_ld_gran4(p0, p1, p2, p3, w0, w1, w2, w3);
        const bool ok = (!h0 || rs_gran_tag(w0) == tag) && (!h1 || rs_gran_tag(w1) == tag) && (!h2 || rs_gran_tag(w2) == tag) && (!h3 || rs_gran_tag(w3) == tag);
        if (__all(ok)) break;
        if ((spins & 4095u) == 0) {      // (rarely: the clock is a scalar memory read, the error word lives in host memory)
          const bool give_up = wall_clock64() - t_start > RS_TIMEOUT_TICKS || __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0;
          if (__any(give_up)) { failed = true; break; }      // (wave-uniform)
        }
        // (a nap between polls - __builtin_amdgcn_s_sleep(1) - measured 11.0 us per iteration against 10.7 without: one wave per workgroup polls, the others wait at the barrier)
      }
      if (failed) break;
      const double g0 = h0 ? rs_gran_val(w0) : 0.0, g1 = h1 ? rs_gran_val(w1) : 0.0, g2 = h2 ? rs_gran_val(w2) : 0.0, g3 = h3 ? rs_gran_val(w3) : 0.0;
      if (mine_max) { acc = g0 > acc ? g0 : acc; acc = g1 > acc ? g1 : acc; acc = g2 > acc ? g2 : acc; acc = g3 > acc ? g3 : acc; }
      else { acc += g0; acc += g1; acc += g2; acc += g3; }
    }
    if (!failed) {
#pragma unroll
      for (int o = 32; o >= NV; o >>= 1) { const double w = __shfl_xor(acc, o, 64); acc = mine_max ? (w > acc ? w : acc) : acc + w; }      // (NV = 2: the lanes of one parity)
      if (lane < NV) s_tot[lane] = acc;
    }
    if (lane == 0) {
      *s_fail = failed ? 1 : 0;
      if (failed) __hip_atomic_store(a.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  __syncthreads();
  if (*s_fail) return false;
#pragma unroll
  for (int n = 0; n < NV; ++n) v[n] = s_tot[n];
  __syncthreads();      // (s_red / s_tot are reused by the next reduction)
  return true;
}

// z = M_tile^-1 r (main.c:602-626 restricted to the tile: k_pcg.hip k_precond_tile) and this wave's share of dot(z, r)
template <typename T>
__device__ __forceinline__ void rs_tile_solve(const T (&rr)[16], const T* ee_l, const unsigned int (&mm)[8], T (&zz)[16], double& dsum) {
  T ee[16];      // E^-1 of the chunk: read-only after the factorisation, kept in LDS between the solves ([record][lane]: conflict-free)
#pragma unroll
  for (int j = 0; j < 16; ++j) ee[j] = ee_l[j * 64];
  T own = (T)-0.0, out = (T)-0.0;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int cm = (int)((mm[j >> 1] >> ((j & 1) * 8)) & 0xff);
    const T nbv = rs_shift<RS_SHR1>(out, (T)-0.0);
    const T t = rr[j] - own - nbv;
    const T qv = t * ee[j];
    const T res = (cm & CM_FLUID) ? qv : (T)0;
    const T carry = (T)-1 * ee[j] * res;
    own = carry; out = carry;
    zz[j] = res;
  }
  own = (T)0; out = (T)0;
#pragma unroll
  for (int j = 15; j >= 0; --j) {
    const int cm = (int)((mm[j >> 1] >> ((j & 1) * 8)) & 0xff);
    const T nbv = rs_shift<RS_SHL1>(out, (T)0);
    const T kr = ((cm & CM_RIGHT) ? (T)-1 : (T)0) * ee[j], ku = ((cm & CM_UP) ? (T)-1 : (T)0) * ee[j];
    const T t = zz[j] - kr * own - ku * nbv;
    const T zv = t * ee[j];
    const T res = (cm & CM_FLUID) ? zv : (T)0;
    own = res; out = res;
    if (cm & CM_FLUID) dsum += (double)res * (double)rr[j];
    zz[j] = res;
  }
}

template <typename T>
__global__ __launch_bounds__(RS_THREADS, 2) void k_pcg_resident(ResArgs a) {      // (<= 256 registers: double 254, float 168 - see the opaque copies in the iteration loop)
  typedef double d2 __attribute__((ext_vector_type(2)));
  PcgScalars* sc = a.sc;
  if (!sc->nonzero) return;      // all_zero(r), main.c:742 (every thread of the grid reads the same word)
  __shared__ double s_red[RS_WAVES][3];
  __shared__ double s_tot[3];
  __shared__ int s_fail;
  __shared__ T s_ee[RS_WAVES][16][64];      // E^-1 and p of the wave's chunk: touched once per iteration, so they live in LDS and leave the registers to r, s, z
  __shared__ T s_pp[RS_WAVES][16][64];
  __shared__ T s_edge[RS_WAVES][32];        // the halo's s' of lane 0 (entries 0..15: the row below the band) and of lane 63 (16..31: the row above), per record
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = (int)sc->n_chunks;
  // The host launches as many workgroups as the device holds at once WITHOUT knowing this solve's number of active chunks (that would be a host round trip per solve):
  // the first (nch + 3) / 4 take part, the others leave at once; a solve that needs more than were launched raises error 2 ("does not fit") and nobody starts.
  const int nwg = (nch + RS_WAVES - 1) / RS_WAVES;
  if (nwg > (int)gridDim.x) { if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(a.err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); return; }
  if (a.force_fail) { if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(a.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); return; }
  if ((int)blockIdx.x >= nwg) return;
  const int ci = blockIdx.x * RS_WAVES + wave;
  const bool act = ci < nch;
  const int TS = a.g.TS, ntb = a.g.T / 16, nbands = a.g.nbands;
  const unsigned int ent = act ? a.list[ci] : 0u;
  const int tile = (int)(ent & ~EU_CHUNK_INTERIOR);
  const int band = a.band_lo + tile / ntb, k = tile % ntb;
  const size_t base = ((size_t)band * TS + (size_t)k * 16) * 64 + 2 * lane;      // element (band, record 16 k, lane)
  size_t vbase = base;      // (the copy the iteration loop uses: see there)
  T* zx = static_cast<T*>(a.zx);
  T* sx = static_cast<T*>(a.sx);

  // ---- the chunk: masks, r = b, E^-1
  unsigned int mm[8];
  T rr[16], ss[16], zz[16];
  T* ee_l = &s_ee[wave][0][lane];
  T* pp_l = &s_pp[wave][0][lane];
  {      // E^-1 per tile (k_factor_tile): a tile starts where a band starts - precon 0 to the left and below; a non-fluid cell keeps its stale entry
    T ee[16];
#pragma unroll
    for (int P = 0; P < 8; ++P) {
      mm[P] = act ? (unsigned int)*reinterpret_cast<const unsigned short*>(a.mask + base + P * 128) : 0u;
      d2 bv = {0.0, 0.0}, ev = {0.0, 0.0};
      if (act) { bv = *reinterpret_cast<const d2*>(a.b + base + P * 128); ev = *reinterpret_cast<const d2*>(a.pre + base + P * 128); }
      rr[2 * P] = (T)bv.x; rr[2 * P + 1] = (T)bv.y;
      ee[2 * P] = (T)ev.x; ee[2 * P + 1] = (T)ev.y;
      ss[2 * P] = ss[2 * P + 1] = (T)0;
      pp_l[(2 * P) * 64] = (T)0; pp_l[(2 * P + 1) * 64] = (T)0;
    }
    T own = (T)0, out = (T)0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int cm = (int)((mm[j >> 1] >> ((j & 1) * 8)) & 0xff);
      const T nbv = rs_shift<RS_SHR1>(out, (T)0);
      const T res = (cm & CM_FLUID) ? rs_factor_step<T>((T)(cm >> CM_DIAG_SHIFT), own, nbv) : ee[j];
      own = res; out = res;
      ee[j] = res;
      ee_l[j * 64] = res;
    }
    if (act) {
#pragma unroll
      for (int P = 0; P < 8; ++P) *reinterpret_cast<d2*>(a.pre + base + P * 128) = d2{(double)ee[2 * P], (double)ee[2 * P + 1]};      // g_precon persists (main.c:577)
    }
  }

  // where this wave's halo lives.  Record 16 k - 1 (odd: the pair before, second slot) and record 16 k + 16; lanes 0..15 of the edge load fetch
  // what lane 0 needs from the band below - (band - 1, lane 63, record 16 k + j + 63) - lanes 16..31 what lane 63 needs from the band above -
  // (band + 1, lane 0, record 16 k + j - 63).  Addresses that do not exist (or are never selected by a mask bit) fall back to the wave's own cell.
  const bool hasL = act && k > 0;
  const long long dL = hasL ? -127 : 0, dR = act ? 1024 : 0;      // (offsets from the chunk's own element: the addresses are formed inside the loop)
  long long dE = 0;
  bool hasE = false;
  if (act && lane < 16 && band > 0) { const int t2 = 16 * k + lane + 63; dE = (long long)(((size_t)(band - 1) * TS + (size_t)(t2 & ~1)) * 64 + 126 + (t2 & 1)) - (long long)base; hasE = true; }
  if (act && lane >= 16 && lane < 32 && band + 1 < nbands) {
    const int t2 = 16 * k + (lane - 16) - 63;
    if (t2 >= 0) { dE = (long long)(((size_t)(band + 1) * TS + (size_t)(t2 & ~1)) * 64 + (t2 & 1)) - (long long)base; hasE = true; }
  }
  auto publish = [&](const T (&zv)[16], const T (&sv)[16]) {      // the cells other waves read: first and last record, lane 0 and lane 63
    if (!act) return;
    T* zb = zx + vbase;
    T* sb = sx + vbase;
    rs_st(zb, zv[0]); rs_st(sb, sv[0]);
    rs_st(zb + 7 * 128 + 1, zv[15]); rs_st(sb + 7 * 128 + 1, sv[15]);
    if (lane == 0 || lane == 63) {
#pragma unroll
      for (int j = 1; j < 15; ++j) { rs_st(zb + (j >> 1) * 128 + (j & 1), zv[j]); rs_st(sb + (j >> 1) * 128 + (j & 1), sv[j]); }
    }
  };

  // ---- z_0 = M^-1 r, sigma = dot(z_0, r) (main.c:745-748)
  unsigned long long tag = a.tag0;
  double sigma;
  {
    double dsum = 0.0;
    if (act) rs_tile_solve<T>(rr, ee_l, mm, zz, dsum);
    publish(zz, ss);
    double v[1] = {dsum};
    const bool mx[1] = {false};
    if (!rs_reduce<1>(a, nwg, v, mx, 0, tag, s_red, &s_tot[0], &s_fail)) return;
    ++tag;
    sigma = v[0];
  }
  double alpha = 0.0, alpha_prev = 0.0, beta = 0.0, rnorm = 0.0, zs = 0.0, sigma_new = 0.0;
  int it = 0, done = 0;
  const int max_it = a.max_iters;
  while (it < max_it) {
    // ---- s' = z + beta s (main.c:669-677; the first search direction is z_0 itself, main.c:746) on the chunk and on its halo, then A s' (main.c:679-691)
    const bool first = it == 0;
    const T bt = (T)beta;
    // Everything derived from the masks and the chunk's address is loop-invariant, and the compiler hoists ALL of it out of the iteration loop - 16 diagonals as
    // doubles, 64-bit addresses of every halo cell ... - ~100 registers of values that cost one or two instructions to recompute.  Opaque copies keep them in the loop:
    // the double kernel then fits 256 registers (two workgroups per CU: twice the chunks) without spilling.
#pragma unroll
    for (int P = 0; P < 8; ++P) asm volatile("" : "+v"(mm[P]));
    asm volatile("" : "+v"(vbase));
    T zL, sL, zR, sR, zE, sE;
    rs_ld6(zx + vbase + dL, sx + vbase + dL, zx + vbase + dR, sx + vbase + dR, zx + vbase + dE, sx + vbase + dE, zL, sL, zR, sR, zE, sE);
    T spL = first ? zL : zL + bt * sL, spR = first ? zR : zR + bt * sR, spE = first ? zE : zE + bt * sE;
    if (!hasL) spL = (T)0;
    if (!hasE) spE = (T)0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int cm = (int)((mm[j >> 1] >> ((j & 1) * 8)) & 0xff);
      if (cm & CM_FLUID) ss[j] = first ? zz[j] : zz[j] + bt * ss[j];      // (a non-fluid cell keeps its +0)
    }
    // lane 0 / lane 63 pick their halo values record by record out of LDS (lanes 0..31 fetched them): the DPP shifts below hand a lane without a
    // source ITS OWN injected register, so one register serves both ends
    if (lane < 32) s_edge[wave][lane] = spE;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (one wave: LDS operations execute in order; this pins the compiler's order)
    const T* edge_l = &s_edge[wave][lane == 63 ? 16 : 0];
    double dsa = 0.0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int cm = (int)((mm[j >> 1] >> ((j & 1) * 8)) & 0xff);
      const T left = j > 0 ? ss[j > 0 ? j - 1 : 0] : spL, right = j < 15 ? ss[j < 15 ? j + 1 : 15] : spR;
      const T ej = edge_l[j];
      const T dn = rs_shift<RS_SHR1>(left, ej), up = rs_shift<RS_SHL1>(right, ej);
      T v = (T)0;
      if (cm & CM_FLUID) {      // diag, right, up, left, down: the reference's order
        v = (T)(int)(cm >> CM_DIAG_SHIFT) * ss[j];
        v = v - ((cm & CM_RIGHT) ? right : (T)0);
        v = v - ((cm & CM_UP) ? up : (T)0);
        v = v - ((cm & CM_LEFT) ? left : (T)0);
        v = v - ((cm & CM_DOWN) ? dn : (T)0);
        dsa += (double)v * (double)ss[j];
      }
      zz[j] = v;      // A s' (z is dead until the next tile solve)
    }
    {
      double v[1] = {dsa};
      const bool mx[1] = {false};
      if (!rs_reduce<1>(a, nwg, v, mx, 0, tag, s_red, &s_tot[0], &s_fail)) return;
      ++tag;
      zs = v[0];
    }
    alpha_prev = alpha; alpha = sigma / zs; it += 1;      // main.c:750-752
    // ---- p += alpha s; r -= alpha A s (main.c:753-754, evaluated as r + (A s) * (-alpha) like k_precond_tile); max |r| (main.c:756)
    const T at = (T)alpha, nat = (T)(-alpha);
    double mxr = 0.0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int cm = (int)((mm[j >> 1] >> ((j & 1) * 8)) & 0xff);
      if (cm & CM_FLUID) {
        pp_l[j * 64] = pp_l[j * 64] + ss[j] * at;
        rr[j] = rr[j] + zz[j] * nat;
        const double av = fabs((double)rr[j]);
        if (av > mxr) mxr = av;
      }
    }
    // ---- z = M^-1 r and dot(z, r) (main.c:760-762) - not on the budget's last iteration, whose result nobody would read
    asm volatile("" ::: "memory");      // (phase boundary: keeps the scheduler from hoisting the next phase's LDS loads into this one - register pressure)
    double dsum = 0.0;
    const bool sweeps = it < max_it;
    if (sweeps) {
      if (act) rs_tile_solve<T>(rr, ee_l, mm, zz, dsum);
      publish(zz, ss);
    }
    {
      double v[2] = {mxr, dsum};
      const bool mx[2] = {true, false};
      if (!rs_reduce<2>(a, nwg, v, mx, 1, tag, s_red, &s_tot[0], &s_fail)) return;
      ++tag;
      rnorm = v[0];
      if (rnorm <= a.tol) done = 1;                                               // main.c:756
      if (!done && sweeps) { sigma_new = v[1]; beta = sigma_new / sigma; sigma = sigma_new; }      // main.c:762-765
    }
    if (done) break;
  }
  // ---- the results: p (clamped later by the velocity update), the final residual, the scalars
  if (act) {
#pragma unroll
    for (int P = 0; P < 8; ++P) {
      *reinterpret_cast<d2*>(a.p + base + P * 128) = d2{(double)pp_l[(2 * P) * 64], (double)pp_l[(2 * P + 1) * 64]};
      *reinterpret_cast<d2*>(a.r + base + P * 128) = d2{(double)rr[2 * P], (double)rr[2 * P + 1]};
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    sc->sigma = sigma; sc->zs = zs; sc->sigma_new = sigma_new; sc->alpha = alpha; sc->alpha_prev = alpha_prev; sc->beta = beta; sc->rnorm = rnorm;
    sc->done = done; sc->iters = it;
  }
}

int g_res_capacity[16][2];             // workgroups resident at once, per device and precision (0 = not asked yet, -1 = cannot be used)
}  // namespace eu_resident
using namespace eu_resident;

// how many workgroups of the resident kernel the device holds at once (0: cannot be used)
int eu_resident_capacity(euler_sim* S, int f32) {
  static int dummy[2];
  int& cap = (S->cfg.device >= 0 && S->cfg.device < 16) ? g_res_capacity[S->cfg.device][f32 ? 1 : 0] : dummy[f32 ? 1 : 0];
  const long long lim = S->opt[EULER_OPT_RESIDENT_CAP];      // (tests: a small capacity, so that a scene outgrows it)
  if (cap != 0) { const int c = cap < 0 ? 0 : cap; return lim > 0 && lim < c ? (int)lim : c; }
  int per_cu = 0, cus = 0;
  hipError_t e = f32 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_pcg_resident<float>, RS_THREADS, 0)
                     : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_pcg_resident<double>, RS_THREADS, 0);
  if (e != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, S->cfg.device) != hipSuccess || per_cu * cus <= 0) { cap = -1; return 0; }
  if (per_cu > 3) per_cu = 3;      // (double: 2 by registers and LDS, float: 3)
  cap = per_cu * cus;
  if (cap > RS_MAX_WG) cap = RS_MAX_WG;
  return lim > 0 && lim < cap ? (int)lim : cap;
}

// launch the whole solve; the caller has run k_pcg_reset + the assembly and knows n_chunks (> 0) and that the right-hand side is not all zero
int eu_launch_resident(euler_sim* S, unsigned int n_chunks) {      // n_chunks: how many workgroups' worth of chunks to launch for (the exact count, or the device's capacity)
  ResArgs a;
  a.g = S->geom; a.mask = S->cellmask; a.b = S->b; a.p = S->p; a.r = S->r; a.pre = S->precon; a.zx = S->z; a.sx = S->s; a.list = S->chunk_list; a.sc = S->sc;
  a.gran = S->res_gran; a.tag0 = S->res_tag; a.band_lo = S->band_lo; a.max_iters = S->cfg.max_iterations; a.tol = S->cfg.tol; a.err = S->res_err;
  S->res_tag += 2ull * (unsigned long long)S->cfg.max_iterations + 4ull;
  a.force_fail = 0;
  if (S->opt[EULER_OPT_RESIDENT_FORCE_TIMEOUT] > 0) { a.force_fail = 1; S->opt[EULER_OPT_RESIDENT_FORCE_TIMEOUT] -= 1; }
  const unsigned nwg = (n_chunks + RS_WAVES - 1) / RS_WAVES;
  if (S->cfg.pcg_precision == EULER_PCG_F32) LAUNCH(S, KC_RESIDENT, k_pcg_resident<float>, dim3(nwg), dim3(RS_THREADS), a);
  else LAUNCH(S, KC_RESIDENT, k_pcg_resident<double>, dim3(nwg), dim3(RS_THREADS), a);
  return EULER_OK;
}
