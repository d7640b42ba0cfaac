// k_markers.hip — marker-particle kernels: advection with solid collision, binning into the
// count grid, order-preserving compaction, fluid sources, and the ordered bit-select primitive
// they share.
//
// The reference's marker array ORDER is observable (see advect below), so everything that
// changes the array (swap-with-last deletion main.c:112, source appends main.c:288) reproduces
// the reference's order exactly, in parallel.
#include "euler_dev.h"

#include <stdlib.h>

#include <float.h>

// ==========================================================================================
// ordered select: indices of the set bits of a bit mask, ascending.  Three launches:
// per-block popcount -> single-workgroup exclusive scan -> per-block emit.
#define SEL_THREADS 256
#define SEL_WPT 8                                 // mask words per thread
#define SEL_WPB (SEL_THREADS * SEL_WPT)           // words per block (131072 items)

__global__ __launch_bounds__(SEL_THREADS) void k_sel_count(const unsigned long long* __restrict__ mask, size_t nwords,
                                                           unsigned int* __restrict__ block_sums) {
  const size_t w0 = (size_t)blockIdx.x * SEL_WPB + (size_t)threadIdx.x * SEL_WPT;
  unsigned int c = 0;
#pragma unroll
  for (int k = 0; k < SEL_WPT; ++k)
    if (w0 + k < nwords) c += __popcll(mask[w0 + k]);
  c = (unsigned int)eu_wave_sum((double)c);   // exact: counts << 2^53
  __shared__ unsigned int sw[SEL_THREADS / 64];
  if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned int t = 0;
    for (int k = 0; k < SEL_THREADS / 64; ++k) t += sw[k];
    block_sums[blockIdx.x] = t;
  }
}

// exclusive scan of block_sums in place (single workgroup, 1024 threads), total -> *total
__global__ __launch_bounds__(1024) void k_sel_scan(unsigned int* block_sums, size_t nblocks, unsigned int* total) {
  __shared__ unsigned int buf[1024];
  __shared__ unsigned int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (size_t base = 0; base < nblocks; base += 1024) {
    const size_t i = base + threadIdx.x;
    const unsigned int v = i < nblocks ? block_sums[i] : 0u;
    buf[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {   // Hillis-Steele inclusive scan
      unsigned int t = threadIdx.x >= (unsigned)o ? buf[threadIdx.x - o] : 0u;
      __syncthreads();
      buf[threadIdx.x] += t;
      __syncthreads();
    }
    const unsigned int incl = buf[threadIdx.x];
    const unsigned int c = carry;
    if (i < nblocks) block_sums[i] = c + incl - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry = c + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = carry;
}

__global__ __launch_bounds__(SEL_THREADS) void k_sel_emit(const unsigned long long* __restrict__ mask, size_t nwords,
                                                          const unsigned int* __restrict__ block_off,
                                                          unsigned int* __restrict__ out, size_t out_cap) {
  const size_t w0 = (size_t)blockIdx.x * SEL_WPB + (size_t)threadIdx.x * SEL_WPT;
  unsigned long long w[SEL_WPT];
  unsigned int c = 0;
#pragma unroll
  for (int k = 0; k < SEL_WPT; ++k) { w[k] = (w0 + k < nwords) ? mask[w0 + k] : 0ull; c += __popcll(w[k]); }
  __shared__ unsigned int sc[SEL_THREADS];
  sc[threadIdx.x] = c;
  __syncthreads();
  for (int o = 1; o < SEL_THREADS; o <<= 1) {
    unsigned int t = threadIdx.x >= (unsigned)o ? sc[threadIdx.x - o] : 0u;
    __syncthreads();
    sc[threadIdx.x] += t;
    __syncthreads();
  }
  size_t off = (size_t)block_off[blockIdx.x] + sc[threadIdx.x] - c;
#pragma unroll
  for (int k = 0; k < SEL_WPT; ++k) {
    unsigned long long m = w[k];
    while (m) {
      const int bit = __ffsll((long long)m) - 1;
      if (off < out_cap) out[off] = (unsigned int)((w0 + k) * 64 + bit);
      ++off;
      m &= m - 1;
    }
  }
}

// The same select as ONE workgroup for a short mask (a small grid's markers, any grid's chunk flags): a thread counts a contiguous slice of words, the 1024
// counts are scanned in LDS, the thread emits its slice's indices behind its offset - the same ascending list, one launch instead of three (1024^2: 19 -> 7 us,
// three times per substep).
#define SEL1_MAX_WORDS (1u << 15)
__device__ __forceinline__ unsigned int sel_wave_excl(unsigned int c, unsigned int* total) {      // exclusive prefix of c over the wave's lanes, and the wave's sum
  unsigned int v = c;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const unsigned int t = __shfl_up(v, o, 64); if ((int)(threadIdx.x & 63) >= o) v += t; }
  *total = __shfl(v, 63, 64);
  return v - c;
}
// wave w takes the words [w seg, (w + 1) seg), 64 at a time with lane l on word base + l (coalesced); the list stays ascending: groups of 64 words in order, lanes in order
__global__ __launch_bounds__(1024) void k_sel_small(const unsigned long long* __restrict__ mask, unsigned int nwords, unsigned int* __restrict__ out, size_t out_cap,
                                                    unsigned int* total) {
  __shared__ unsigned int wsum[16];
  const unsigned int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const unsigned int seg = ((nwords + 15u) / 16u + 63u) & ~63u, w0 = wave * seg, w1 = w0 + seg < nwords ? w0 + seg : nwords;
  unsigned int c = 0;
  for (unsigned int w = w0 + lane; w < w1; w += 64) c += __popcll(mask[w]);
  unsigned int tot;
  (void)sel_wave_excl(c, &tot);
  if (lane == 0) wsum[wave] = tot;
  __syncthreads();
  size_t off = 0;
  for (unsigned int k = 0; k < wave; ++k) off += wsum[k];
  if (threadIdx.x == 0) { unsigned int t = 0; for (int k = 0; k < 16; ++k) t += wsum[k]; *total = t; }
  for (unsigned int base = w0; base < w1; base += 64) {      // (uniform per wave)
    const unsigned int w = base + lane;
    unsigned long long m = w < w1 ? mask[w] : 0ull;
    unsigned int gt;
    size_t o = off + sel_wave_excl((unsigned int)__popcll(m), &gt);
    while (m) {
      const int bit = __ffsll((long long)m) - 1;
      if (o < out_cap) out[o] = w * 64u + (unsigned int)bit;
      ++o;
      m &= m - 1;
    }
    off += gt;
  }
}

int eu_ordered_select(euler_sim* S, const unsigned long long* mask, size_t nwords, unsigned int* out_idx,
                      unsigned int* out_total) {
  if (nwords <= SEL1_MAX_WORDS) {
    LAUNCH(S, KC_SELECT, k_sel_small, dim3(1), dim3(1024), mask, (unsigned int)nwords, out_idx, S->sel_cap, out_total);
    return EULER_OK;
  }
  const size_t nblocks = (nwords + SEL_WPB - 1) / SEL_WPB;
  if (nblocks > S->sel.capacity_blocks) { eu_set_error("ordered select: %zu blocks > capacity", nblocks); return EULER_EINVAL; }
  const unsigned nb = nblocks ? (unsigned)nblocks : 1u;
  LAUNCH(S, KC_SELECT, k_sel_count, dim3(nb), dim3(SEL_THREADS), mask, nwords, S->sel.block_sums);
  LAUNCH(S, KC_SELECT, k_sel_scan, dim3(1), dim3(1024), S->sel.block_sums, nblocks ? nblocks : (size_t)1, out_total);
  LAUNCH(S, KC_SELECT, k_sel_emit, dim3(nb), dim3(SEL_THREADS), mask, nwords, S->sel.block_sums, out_idx, S->sel_cap);
  return EULER_OK;
}

// ==========================================================================================
// advect_markers (main.c:464-537).
//
// The reference walks the marker array sequentially and, when a marker hits a solid after
// having crossed a cell boundary, executes `dt -= t_prev` on the FUNCTION PARAMETER (main.c:501,
// 518): every later marker in the array moves with the shortened dt.  Reproduced in parallel:
//   pass A  every marker moves with the incoming dt (speculation); a marker that would shorten dt
//           records (theta = time of the colliding crossing, delta = t_prev) and votes in a ballot;
//   select  the voters' indices in array order;
//   walk    one wave replays the chain in order: if (theta < dt) dt -= delta  -> list of
//           (index, dt after it) for the collisions that really fire;
//   pass B  markers behind the first firing collision are recomputed from their old position
//           with the dt valid at their index.
// Under the CFL bound (displacement <= 0.75 cell, main.c:838) a marker crosses at most one
// boundary per axis, so it can shorten dt at most once; `multi_events` counts violations.
struct AdvectOut { float px, py, theta, delta; int events; };

__device__ __forceinline__ float time_to(float p0, float p1, float vel) {   // main.c:451-457
  return fabsf(vel) > 0.f ? (p1 - p0) / vel : FLT_MAX;
}

// TR: u, v, g.count and solid are the COLUMN-major copies (k_transpose_for_markers): the marker array walks the grid column by column (main.c:243-266), so a wave's
// 64 markers sit in ~16 cells of one column - 16-17 cache lines per gather in the row-major fields, one or two in the column-major ones.  The same values: the same bits.
template <bool TR>
__device__ __forceinline__ AdvectOut advect_one(const GridRef& g, const float* __restrict__ u, const float* __restrict__ v,
                                                const uint8_t* __restrict__ solid, float px, float py, float dt) {
  AdvectOut o;
  o.events = 0; o.theta = 0.f; o.delta = 0.f;
  // velocity_at (main.c:440-449)
  float vx = eu_interp<1, TR>(g, u, px / EU_H - 1.f, py / EU_H - 0.5f);
  float vy = eu_interp<2, TR>(g, v, px / EU_H - 0.5f, py / EU_H - 1.f);
  auto is_solid = [&](int sy_, int sx_) -> bool { return solid[TR ? (size_t)sx_ * g.Y + sy_ : (size_t)sy_ * g.X + sx_] != 0; };
  int xi = (int)floorf(px / EU_H), yi = (int)floorf(py / EU_H);
  const int xdir = vx > 0 ? 1 : -1;
  int nxi = xi + (vx > 0 ? 1 : 0);
  float npx = nxi * EU_H;
  const int ydir = vy > 0 ? 1 : -1;
  int nyi = yi + (vy > 0 ? 1 : 0);
  float npy = nyi * EU_H;
  float tx = time_to(px, npx, vx);
  const int xoff = vx < 0 ? -1 : 0;
  float ty = time_to(py, npy, vy);
  const int yoff = vy < 0 ? -1 : 0;
  float t_prev = 0.f, t_near = fminf(tx, ty);
  int guard = 0;
  while (t_near < dt && guard++ < 64) {
    if (tx < ty) {
      if (is_solid(yi, nxi + xoff)) {
        if (t_prev > 0.f) { if (o.events == 0) { o.theta = t_near; o.delta = t_prev; } o.events++; }
        px = px + t_prev * vx; py = py + t_prev * vy;
        dt -= t_prev; t_near = 0.f; vx = 0.f; tx = FLT_MAX;
        ty = time_to(py, npy, vy);
      } else {
        xi = nxi; nxi = xi + xdir; npx = nxi * EU_H;
        tx = time_to(px, npx, vx);
      }
    } else {
      if (is_solid(nyi + yoff, xi)) {
        if (t_prev > 0.f) { if (o.events == 0) { o.theta = t_near; o.delta = t_prev; } o.events++; }
        px = px + t_prev * vx; py = py + t_prev * vy;
        dt -= t_prev; t_near = 0.f; vy = 0.f; ty = FLT_MAX;
        tx = time_to(px, npx, vx);
      } else {
        yi = nyi; nyi = yi + ydir; npy = nyi * EU_H;
        ty = time_to(py, npy, vy);
      }
    }
    t_prev = t_near;
    t_near = fminf(tx, ty);
  }
  const float t = (t_near < FLT_MAX) ? dt : t_prev;
  o.px = px + t * vx;
  o.py = py + t * vy;
  return o;
}

template <bool TR>
__global__ __launch_bounds__(256) void k_advect_markers_a(const float2* __restrict__ in, float2* __restrict__ out,
                                                          const float* __restrict__ u, const float* __restrict__ v,
                                                          const uint8_t* __restrict__ solid, GridRef g, float dt,
                                                          unsigned long long n, unsigned long long* __restrict__ evmask,
                                                          float* __restrict__ ev_theta, float* __restrict__ ev_delta,
                                                          MarkerState* ms) {
  const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  bool ev = false;
  if (i < n) {
    const float2 p = in[i];
    const AdvectOut o = advect_one<TR>(g, u, v, solid, p.x, p.y, dt);
    out[i] = make_float2(o.px, o.py);
    if (o.events) {
      ev = true;
      ev_theta[i] = o.theta;
      ev_delta[i] = o.delta;
      if (o.events > 1) atomicAdd(&ms->multi_events, 1ull);
    }
  }
  const unsigned long long b = __ballot(ev);
  if ((threadIdx.x & 63) == 0 && (i >> 6) < ((n + 63) >> 6)) evmask[i >> 6] = b;
}

// Two markers per thread (the column-major path of large grids): the kernel is bound by the latency of its dependent chain - marker, indices, gathers, store - at full
// occupancy (1.13 ms for 134 M markers: 1.9 TB/s, whether the gathers are 20 loads or 11, whether time_to divides or not), so every thread runs two independent chains.
// Markers 2 t and 2 t + 1: one 16-byte load and store per thread; evmask keeps its meaning (bit b of word w = marker 64 w + b): a wave covers two words.
__device__ __forceinline__ unsigned long long mk_spread32(unsigned long long x) {      // bit k -> bit 2 k
  x &= 0xffffffffull;
  x = (x | (x << 16)) & 0x0000ffff0000ffffull; x = (x | (x << 8)) & 0x00ff00ff00ff00ffull; x = (x | (x << 4)) & 0x0f0f0f0f0f0f0f0full;
  x = (x | (x << 2)) & 0x3333333333333333ull; x = (x | (x << 1)) & 0x5555555555555555ull;
  return x;
}
template <bool TR>
__global__ __launch_bounds__(256) void k_advect_markers_a2(const float2* __restrict__ in, float2* __restrict__ out,
                                                           const float* __restrict__ u, const float* __restrict__ v,
                                                           const uint8_t* __restrict__ solid, GridRef g, float dt,
                                                           unsigned long long n, unsigned long long* __restrict__ evmask,
                                                           float* __restrict__ ev_theta, float* __restrict__ ev_delta,
                                                           MarkerState* ms) {
  const unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x, i = 2 * t;
  bool ev0 = false, ev1 = false;
  if (i + 1 < n) {
    const float4 p = *reinterpret_cast<const float4*>(in + i);
    const AdvectOut o0 = advect_one<TR>(g, u, v, solid, p.x, p.y, dt);
    const AdvectOut o1 = advect_one<TR>(g, u, v, solid, p.z, p.w, dt);
    *reinterpret_cast<float4*>(out + i) = make_float4(o0.px, o0.py, o1.px, o1.py);
    if (o0.events) { ev0 = true; ev_theta[i] = o0.theta; ev_delta[i] = o0.delta; if (o0.events > 1) atomicAdd(&ms->multi_events, 1ull); }
    if (o1.events) { ev1 = true; ev_theta[i + 1] = o1.theta; ev_delta[i + 1] = o1.delta; if (o1.events > 1) atomicAdd(&ms->multi_events, 1ull); }
  } else if (i < n) {
    const float2 p = in[i];
    const AdvectOut o0 = advect_one<TR>(g, u, v, solid, p.x, p.y, dt);
    out[i] = make_float2(o0.px, o0.py);
    if (o0.events) { ev0 = true; ev_theta[i] = o0.theta; ev_delta[i] = o0.delta; if (o0.events > 1) atomicAdd(&ms->multi_events, 1ull); }
  }
  const unsigned long long b0 = __ballot(ev0), b1 = __ballot(ev1);
  if ((threadIdx.x & 63) == 0) {
    const unsigned long long w = i >> 6, nwords = (n + 63) >> 6;      // (i = the wave's first marker: a multiple of 128)
    if (w < nwords) evmask[w] = mk_spread32(b0) | (mk_spread32(b1) << 1);
    if (w + 1 < nwords) evmask[w + 1] = mk_spread32(b0 >> 32) | (mk_spread32(b1 >> 32) << 1);
  }
}

// ------------------------------------------------------------------------------------------ round 6: advection and binning in ONE pass
// refresh_marker_counts (main.c:102-117) bins the positions advect_markers (main.c:464-537) has just written.  As two kernels that is 8 B per marker read a second time and a
// pass whose time is the memory-side adds of the counters, while the advection pass is bound by its arithmetic and gathers: the two do not compete.  Here the speculative pass
// bins what it has computed - cell, sink | solid test (the delete ballot), the run-aggregated adds - and pass B, which recomputes the markers behind a firing dt collision,
// moves the counts of those whose cell changed (k_advect_markers_b<TR, true>).  Integer adds: the counters come out the same whatever the order.
// A thread holds TWO markers: their cells mostly coincide (markers are seeded four to a cell, consecutively), so the thread's item is (cell, weight 1 or 2) and a run of lanes
// adds the sum of its weights; the rare thread with two different cells bins the second one in a second, wave-uniform round.
__device__ __forceinline__ void bin_aggregated_w(unsigned int* count32, bool live, size_t c, unsigned int w) {      // bin_aggregated (euler_dev.h) with a weight of 1 or 2 per lane
  const unsigned int lo = (unsigned int)c, hi = (unsigned int)(c >> 32);
  const int lane = threadIdx.x & 63;
  const unsigned int plo = __shfl_up(lo, 1, 64), phi = __shfl_up(hi, 1, 64);
  const bool plive = __shfl_up((int)live, 1, 64) != 0;
  const bool head = live && (lane == 0 || !plive || plo != lo || phi != hi);     // first lane of a run of equal cells
  const unsigned long long heads = __ballot(head), lives = __ballot(live), twos = __ballot(live && w == 2u);
  const unsigned long long above = lane == 63 ? 0ull : ((heads | ~lives) >> (lane + 1));
  const int run = above ? __ffsll((long long)above) : 64 - lane;                 // lanes in the run
  const unsigned long long rmask = (run >= 64 ? ~0ull : ((1ull << run) - 1ull)) << lane;
  const int wsum = run + __popcll(twos & rmask);                                 // markers in the run
  const int nl = lane + run;
  const bool has_next = head && nl < 64 && ((heads >> nl) & 1ull);
  const int src = has_next ? nl : lane;
  const unsigned int nlo = __shfl(lo, src, 64), nhi = __shfl(hi, src, 64);
  const int nsum = __shfl(wsum, src, 64);
  const bool pairs = has_next && !(lo & 1u) && nlo == lo + 1u && nhi == hi;      // (c even: the 64-bit add is aligned)
  const unsigned long long below = heads & ((1ull << lane) - 1ull);
  const int pl = below ? 63 - __clzll((long long)below) : lane;
  const unsigned int qlo = __shfl(lo, pl, 64), qhi = __shfl(hi, pl, 64);
  const int qrun = __shfl(run, pl, 64);
  const bool absorbed = head && below && (lo & 1u) && qlo + 1u == lo && qhi == hi && pl + qrun == lane;
  if (head && !absorbed) {
    if (pairs) atomicAdd(reinterpret_cast<unsigned long long*>(&count32[c]), (unsigned long long)(unsigned int)wsum | ((unsigned long long)(unsigned int)nsum << 32));
    else atomicAdd(&count32[c], (unsigned int)wsum);
  }
}
__device__ __forceinline__ size_t mk_cell_colmajor(float px, float py, int H) {      // the counters' (and blockedT's) index of the cell a position lies in
  return (size_t)(int)floorf(px / EU_H) * H + (int)floorf(py / EU_H);
}
// the tile map's third array (euler_sim.tmap + 2 tmap_n): "a marker ENTERED this tile since the last refresh".  A marker that stays inside its tile needs no mark - the tile
// held it at the last refresh, so the count grid's flag is up - and k_narrow_counts<true> reads the counters of a tile only if one of the two says so.
__device__ __forceinline__ int mk_tile(float px, float py, int tnx) { return ((int)floorf(py / EU_H) >> 6) * tnx + ((int)floorf(px / EU_H) >> 6); }
// the counters' index of the cell the NEW position lies in (mk_cell_colmajor), marking its tile when the OLD position lay in another one (positions are never negative:
// truncation is floor; EU_H = 1)
__device__ __forceinline__ size_t mk_cell_touch(float ox, float oy, float px, float py, int H, bool live, uint8_t* touch, int tnx) {
  const int nx = (int)floorf(px / EU_H), ny = (int)floorf(py / EU_H);
  if (live && (((nx ^ (int)ox) | (ny ^ (int)oy)) >> 6) != 0) touch[(ny >> 6) * tnx + (nx >> 6)] = 1;      // (16384^2 dam break: + 55 us here, - 170 us in k_narrow_counts)
  return (size_t)nx * H + ny;
}
template <bool TR>
__global__ __launch_bounds__(256) void k_advect_bin_a2(const float2* __restrict__ in, float2* __restrict__ out,
                                                       const float* __restrict__ u, const float* __restrict__ v,
                                                       const uint8_t* __restrict__ solid, GridRef g, float dt,
                                                       unsigned long long n, unsigned long long* __restrict__ evmask,
                                                       float* __restrict__ ev_theta, float* __restrict__ ev_delta,
                                                       MarkerState* ms, const uint8_t* __restrict__ blockedT, unsigned int* count32,
                                                       unsigned long long* __restrict__ delmask, int H, uint8_t* touch, int tnx) {
  const unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x, i = 2 * t;
  bool ev0 = false, ev1 = false, del0 = false, del1 = false, live0 = false, live1 = false;
  size_t c0 = 0, c1 = 0;
  if (i + 1 < n) {
    const float4 p = *reinterpret_cast<const float4*>(in + i);
    const AdvectOut o0 = advect_one<TR>(g, u, v, solid, p.x, p.y, dt);
    const AdvectOut o1 = advect_one<TR>(g, u, v, solid, p.z, p.w, dt);
    *reinterpret_cast<float4*>(out + i) = make_float4(o0.px, o0.py, o1.px, o1.py);
    if (o0.events) { ev0 = true; ev_theta[i] = o0.theta; ev_delta[i] = o0.delta; if (o0.events > 1) atomicAdd(&ms->multi_events, 1ull); }
    if (o1.events) { ev1 = true; ev_theta[i + 1] = o1.theta; ev_delta[i + 1] = o1.delta; if (o1.events > 1) atomicAdd(&ms->multi_events, 1ull); }
    c0 = mk_cell_touch(p.x, p.y, o0.px, o0.py, H, true, touch, tnx); c1 = mk_cell_touch(p.z, p.w, o1.px, o1.py, H, true, touch, tnx);      // (a marker about to be deleted marks too: a superset)
    del0 = blockedT[c0] != 0; del1 = blockedT[c1] != 0;      // refresh_marker_counts drops markers in sink or solid cells (main.c:109-112)
    live0 = !del0; live1 = !del1;
  } else if (i < n) {
    const float2 p = in[i];
    const AdvectOut o0 = advect_one<TR>(g, u, v, solid, p.x, p.y, dt);
    out[i] = make_float2(o0.px, o0.py);
    if (o0.events) { ev0 = true; ev_theta[i] = o0.theta; ev_delta[i] = o0.delta; if (o0.events > 1) atomicAdd(&ms->multi_events, 1ull); }
    c0 = mk_cell_touch(p.x, p.y, o0.px, o0.py, H, true, touch, tnx);
    del0 = blockedT[c0] != 0;
    live0 = !del0;
  }
  const bool same = live0 && live1 && c0 == c1;
  bin_aggregated_w(count32, live0 || live1, live0 ? c0 : c1, same ? 2u : 1u);
  const bool second = live0 && live1 && c0 != c1;
  if (__any(second)) bin_aggregated(count32, second, c1);
  const unsigned long long b0 = __ballot(ev0), b1 = __ballot(ev1), d0 = __ballot(del0), d1 = __ballot(del1);
  if ((threadIdx.x & 63) == 0) {
    const unsigned long long w = i >> 6, nwords = (n + 63) >> 6;      // (i = the wave's first marker: a multiple of 128)
    if (w < nwords) { evmask[w] = mk_spread32(b0) | (mk_spread32(b1) << 1); delmask[w] = mk_spread32(d0) | (mk_spread32(d1) << 1); }
    if (w + 1 < nwords) { evmask[w + 1] = mk_spread32(b0 >> 32) | (mk_spread32(b1 >> 32) << 1); delmask[w + 1] = mk_spread32(d0 >> 32) | (mk_spread32(d1 >> 32) << 1); }
  }
}

// one wave: replay the dt chain over the candidate collisions in array order
__global__ __launch_bounds__(64) void k_marker_walk(const unsigned int* __restrict__ ev_idx, const float* __restrict__ ev_theta,
                                                    const float* __restrict__ ev_delta, unsigned int* __restrict__ act_idx,
                                                    float* __restrict__ act_dt, MarkerState* ms, float dt0) {
  const unsigned int K = ms->n_events;
  const int lane = threadIdx.x;
  float dt = dt0;
  unsigned int M = 0;
  for (unsigned int base = 0; base < K; base += 64) {
    unsigned int idx = 0; float th = 0.f, de = 0.f;
    if (base + lane < K) { idx = ev_idx[base + lane]; th = ev_theta[idx]; de = ev_delta[idx]; }
    const unsigned int cnt = K - base < 64u ? K - base : 64u;
    for (unsigned int j = 0; j < cnt; ++j) {
      const float thj = __shfl(th, (int)j, 64), dej = __shfl(de, (int)j, 64);
      const unsigned int ij = __shfl(idx, (int)j, 64);
      if (thj < dt) {
        dt = dt - dej;
        if (lane == 0) { act_idx[M] = ij; act_dt[M] = dt; }
        ++M;
      }
    }
  }
  if (lane == 0) { ms->n_actual = M; ms->dt_final = dt; ms->total_dt_events += M; }
}

// FIX (behind k_advect_bin_a2): a recomputed marker whose cell changed takes its count along, and the delete ballot of its word is made current.
// A workgroup walks the array with a stride (the launch leaves at once when no collision fired: half a million workgroups took 110 us to do so at 8192^2).
template <bool TR, bool FIX>
__global__ __launch_bounds__(256) void k_advect_markers_b(const float2* __restrict__ in, float2* __restrict__ out,
                                                          const float* __restrict__ u, const float* __restrict__ v,
                                                          const uint8_t* __restrict__ solid, GridRef g,
                                                          unsigned long long n, const unsigned int* __restrict__ act_idx,
                                                          const float* __restrict__ act_dt, const MarkerState* ms,
                                                          const unsigned int* __restrict__ keys,      // keys: slab mode, the markers' GLOBAL array indices
                                                          const uint8_t* __restrict__ blockedT, unsigned int* count32, unsigned long long* delmask, int H,
                                                          uint8_t* touch, int tnx) {
  const unsigned int M = ms->n_actual;
  if (M == 0) return;
  const unsigned int first = act_idx[0];
  for (unsigned long long base = (unsigned long long)blockIdx.x * blockDim.x; base < n; base += (unsigned long long)gridDim.x * blockDim.x) {      // (uniform per workgroup)
    const unsigned long long i = base + threadIdx.x;
    bool redo = false, ndel = false;
    if (i < n) {
      const unsigned long long gi = keys ? keys[i] : i;      // the position in the reference's array decides which dt applies
      if (gi > first) {
        // number of firing collisions with index < gi  (act_idx ascending)
        unsigned int lo = 0, hi = M;
        while (lo < hi) { const unsigned int mid = (lo + hi) >> 1; if (act_idx[mid] < gi) lo = mid + 1; else hi = mid; }
        const float dt = act_dt[lo - 1];
        const float2 p = in[i];
        const AdvectOut o = advect_one<TR>(g, u, v, solid, p.x, p.y, dt);
        if (FIX) {
          const float2 was = out[i];      // pass A's result: binned already
          const size_t oc = mk_cell_colmajor(was.x, was.y, H), nc = mk_cell_colmajor(o.px, o.py, H);
          redo = true;
          ndel = blockedT[nc] != 0;
          if (oc != nc) {
            if (blockedT[oc] == 0) atomicSub(&count32[oc], 1u);
            if (!ndel) { atomicAdd(&count32[nc], 1u); touch[mk_tile(o.px, o.py, tnx)] = 1; }      // (the tile map: the marker may have entered the tile)
          }
        }
        out[i] = make_float2(o.px, o.py);
      }
    }
    if (FIX) {
      const unsigned long long rb = __ballot(redo), nb = __ballot(redo && ndel);
      if ((threadIdx.x & 63) == 0 && rb) { const unsigned long long w = i >> 6; delmask[w] = (delmask[w] & ~rb) | nb; }      // (i: the wave's first marker)
    }
  }
}

// the marker stage's column-major copies: u, v and the count grid as they stand in front of advect_markers (main.c:855), the solid grid when it changed.  A workgroup moves
// a 64 x 64 tile through LDS: 256-byte rows in, 256-byte columns out.  9 bytes per cell read and written per substep - a quarter of a millisecond at 8192^2 against the
// millisecond the advection saves.
#define TM_ROWS 32      // a workgroup's tile: 64 columns x TM_ROWS rows - 21 KB of LDS, seven workgroups per CU (64 x 64: 42 KB, three: the pass was latency-bound)
// countT does not hold counts (round 6): per cell, the typed fluid properties of the FOUR CORNERS of an interpolation whose base cell it is (eu_interp<.., TR>) -
// bits 0-3 for a U-typed field (v00 = c(x,y) | c(x+1,y), v01 = c(x+1,y) | c(x+2,y), v10, v11 the same one row up), bits 4-7 for a V-typed one
// (v00 = c(x,y) | c(x,y+1), v01 = c(x+1,y) | c(x+1,y+1), v10 = c(x,y+1) | c(x,y+2), v11 = c(x+1,y+1) | c(x+1,y+2)); counts beyond the grid read as 0 (the
// interpolation's clamps never look there).  One byte gather per interpolation instead of six: the advection pass is bound by the number of its gather
// instructions and the lines each touches (timing experiments, profiles/r06_marker_gather_experiment.md: 1078 -> 683 us without the twelve count loads of a thread).
__global__ __launch_bounds__(256) void k_transpose_for_markers(const float* __restrict__ u, const float* __restrict__ v, const uint8_t* __restrict__ count, const uint8_t* __restrict__ solid,
                                                               float* __restrict__ uT, float* __restrict__ vT, uint8_t* __restrict__ countT, uint8_t* __restrict__ solidT,
                                                               int X, int Y, int with_solid, const uint8_t* __restrict__ tmap, int tnx, int tn) {
  // round 6 (tmap: countT is what this kernel left one refresh ago): no water in or next to the tile in either count grid - countT holds the zeros it would get, u and v stay
  const int bx = (int)blockIdx.x, by = (int)blockIdx.y;
  if (tmap && eu_tiles_idle(tmap, tnx, tn, (by * TM_ROWS) >> 6, bx, 1)) return;
  __shared__ float tu[TM_ROWS][65], tv[TM_ROWS][65];
  __shared__ uint8_t ts[TM_ROWS][65];
  __shared__ uint8_t tw[TM_ROWS + 2][68];      // per cell of the tile and of the two rows above it: bits 0-2 = "cell x / x + 1 / x + 2 of this row holds markers" (columns of the tile only)
  __shared__ uint8_t e_hi[TM_ROWS + 2][2];     // ... and the same for the two columns to the right of the tile (xb + 64, xb + 65)
  __shared__ int s_any;
  const int xb = bx * 64, yb = by * TM_ROWS, l = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (threadIdx.x == 0) s_any = 0;
  __syncthreads();
  // Round 6: a sample of u or v is only ever USED where its typed fluid property holds (eu_interp selects the others away: main.c:348-362) - a U sample next to a cell
  // with markers, a V sample below or above one.  A tile whose cells, the column to its right and the row above it hold no marker has no such sample: its u and v are
  // not moved (the copies keep what an earlier substep left; nothing reads it).  The half tank's air, most of a dam break's grid.
  bool any = false;
  {
    // rows yb + k, k = w, w + 4, ...: the wave's lanes on columns xb + l; every load first, then a ballot per row (uniform over the wave: a lane's three bits are a shift away).
    // The two columns to the right: ONE load by the first 2 (TM_ROWS + 2) threads (a masked load per row and wave cost 56 us at 8192^2: the pass is bound by its memory instructions)
    constexpr int NIT = (TM_ROWS + 2 + 3) / 4;
    uint8_t cv[NIT], sv[NIT];
#pragma unroll
    for (int q = 0; q < NIT; ++q) {
      const int k = w + 4 * q, y = yb + k, x = xb + l;
      const bool row = k < TM_ROWS + 2 && y < Y;
      cv[q] = (row && x < X) ? count[(size_t)y * X + x] : (uint8_t)0;
      sv[q] = (with_solid && k < TM_ROWS && row && x < X) ? solid[(size_t)y * X + x] : (uint8_t)0;
    }
    if (threadIdx.x < 2 * (TM_ROWS + 2)) {
      const int k = threadIdx.x >> 1, j = threadIdx.x & 1, x = xb + 64 + j, y = yb + k;
      const bool c = x < X && y < Y && count[(size_t)y * X + x] != 0;
      e_hi[k][j] = c ? 1 : 0;
      if (c && j == 0 && k < TM_ROWS) s_any = 1;      // the column to the right of the tile
    }
#pragma unroll
    for (int q = 0; q < NIT; ++q) {
      const int k = w + 4 * q;
      const unsigned long long b = __ballot(cv[q] != 0);
      if (k < TM_ROWS + 2) {      // (uniform)
        tw[k][l] = (uint8_t)((b >> l) & 7ull);
        any = any || (k <= TM_ROWS && b != 0ull);      // the tile and the row above it
        if (with_solid && k < TM_ROWS) ts[k][l] = sv[q];
      }
    }
  }
  if (any && l == 0) s_any = 1;
  __syncthreads();
  const bool move = s_any != 0;
  if (move)
    for (int k = w; k < TM_ROWS; k += 4) {
      const int x = xb + l, y = yb + k;
      const bool in = x < X && y < Y;
      const size_t i = (size_t)y * X + x;
      tu[k][l] = in ? u[i] : 0.f; tv[k][l] = in ? v[i] : 0.f;
    }
  __syncthreads();
  const int r = l % TM_ROWS, h = l / TM_ROWS;              // a wave writes 64 / TM_ROWS columns at a time, TM_ROWS consecutive rows of each
  for (int k = w * (64 / TM_ROWS) + h; k < 64; k += 4 * (64 / TM_ROWS)) {      // column xb + k, rows yb + r
    const int x = xb + k, y = yb + r;
    if (x < X && y < Y) {
      const size_t i = (size_t)x * Y + y;
      if (move) { uT[i] = tu[r][k]; vT[i] = tv[r][k]; }
      // the cells (x .. x + 2) of the rows y, y + 1, y + 2 as three bits each (the last two columns of the tile look beyond it)
      auto win = [&](int row) -> unsigned int {
        unsigned int t = tw[row][k];
        if (k >= 62) t |= k == 62 ? (unsigned int)e_hi[row][0] << 2 : ((unsigned int)e_hi[row][0] << 1) | ((unsigned int)e_hi[row][1] << 2);
        return t;
      };
      const unsigned int A = win(r), B = win(r + 1), C = win(r + 2);
      // U-typed corners: v00 = c(x,y) | c(x+1,y), v01 = c(x+1,y) | c(x+2,y), v10 / v11 one row up; V-typed: v00 = c(x,y) | c(x,y+1), v01 = c(x+1,y) | c(x+1,y+1), v10 / v11 one row up
      countT[i] = (uint8_t)(((A | (A >> 1)) & 3u) | (((B | (B >> 1)) & 3u) << 2) | (((A | B) & 3u) << 4) | (((B | C) & 3u) << 6));
      if (with_solid) solidT[i] = ts[r][k];
    }
  }
}
static bool eu_markers_column_major(euler_sim* S) {      // (whole-grid handles; EULER_OPT_MARKERS_ROWMAJOR: the row-major kernels, for A-B timing)
  const bool off = S->opt[EULER_OPT_MARKERS_ROWMAJOR] != 0;
  if (off || S->slab_on || !S->uT) return false;
  const bool lean = eu_tile_map_on(S) && S->countT_clean == 2 && !S->solidT_dirty;
  LAUNCH(S, KC_MARKER_ADVECT, k_transpose_for_markers, dim3((S->X + 63) / 64, (S->Y + TM_ROWS - 1) / TM_ROWS), dim3(256), S->u, S->v, S->count, S->solid, S->uT, S->vT, S->countT, S->solidT,
         S->X, S->Y, S->solidT_dirty, lean ? (const uint8_t*)S->tmap : (const uint8_t*)nullptr, S->tmap_nx, S->tmap_n);
  S->solidT_dirty = 0;
  S->countT_clean = 1;
  return true;
}

__global__ void k_blocked_transpose(const uint8_t* __restrict__ sink, const uint8_t* __restrict__ solid, uint8_t* __restrict__ blockedT, int X, int Y);
static int eu_blocked_current(euler_sim* S) {      // sink | solid, column-major (the solid / sink grids changed: scenario load, euler_set_field, a snapshot)
  if (S->blocked_dirty) {
    LAUNCH(S, KC_MARKER_BIN, k_blocked_transpose, dim3((S->X + 63) / 64, (S->Y + 63) / 64), dim3(256), S->sink, S->solid, S->blockedT, S->X, S->Y);
    S->blocked_dirty = 0;
  }
  return EULER_OK;
}
// the binning counters are all zero between two refreshes (k_narrow_counts<true> clears what it reads); a refresh that did not get that far leaves them marked
static int eu_count32_clean(euler_sim* S) {
  if (S->count32_dirty) HIPCHK(hipMemsetAsync(S->count32, 0, S->Cw * sizeof(unsigned int), S->stream));
  S->count32_dirty = 1;      // (from here until the next k_narrow_counts<true>)
  return EULER_OK;
}

int eu_launch_advect_markers(euler_sim* S, float dt) {
  const unsigned long long n = S->n_markers_host;
  GridRef g{S->X, S->Y, S->count, S->interp_lim[0], S->interp_lim[1], S->interp_lim[2], S->interp_lim[3]};
  const float2* in = S->markers[S->cur];
  float2* out = S->markers[S->cur ^ 1];
  const unsigned nb = eu_blocks((size_t)n, 256), nb_b = eu_blocks((size_t)n, 256, 4096);
  S->prebin_valid = 0;
  if (eu_markers_column_major(S)) {
    GridRef gt = g;
    gt.count = S->countT;
    const bool fuse = S->opt[EULER_OPT_MARKERS_TWO_PASS] == 0 && S->delmask != nullptr;      // (round 6: the speculative pass bins as well; EULER_OPT_MARKERS_TWO_PASS: rounds 4-5's form)
    if (fuse) {
      int rc = eu_blocked_current(S);
      if (!rc) rc = eu_count32_clean(S);
      if (rc) return rc;
      LAUNCH(S, KC_MARKER_ADVECT, k_advect_bin_a2<true>, dim3(eu_blocks((size_t)((n + 1) / 2), 256)), dim3(256), in, out, S->uT, S->vT, S->solidT, gt, dt, n, S->evmask, S->ev_theta, S->ev_delta, S->ms,
             S->blockedT, S->count32, S->delmask, S->Y, S->tmap + 2 * (size_t)S->tmap_n, S->tmap_nx);
    } else
      LAUNCH(S, KC_MARKER_ADVECT, k_advect_markers_a2<true>, dim3(eu_blocks((size_t)((n + 1) / 2), 256)), dim3(256), in, out, S->uT, S->vT, S->solidT, gt, dt, n, S->evmask, S->ev_theta, S->ev_delta, S->ms);
    int rc = eu_ordered_select(S, S->evmask, (size_t)((n + 63) / 64), S->sel_idx, &S->ms->n_events);
    if (rc) return rc;
    LAUNCH(S, KC_MARKER_EVENTS, k_marker_walk, dim3(1), dim3(64), S->sel_idx, S->ev_theta, S->ev_delta, S->act_idx, S->act_dt, S->ms, dt);
    if (fuse) {
      LAUNCH(S, KC_MARKER_ADVECT, (k_advect_markers_b<true, true>), dim3(nb_b), dim3(256), in, out, S->uT, S->vT, S->solidT, gt, n, S->act_idx, S->act_dt, S->ms, (const unsigned int*)nullptr,
             S->blockedT, S->count32, S->delmask, S->Y, S->tmap + 2 * (size_t)S->tmap_n, S->tmap_nx);
      S->prebin_valid = 1;      // (consumed by the refresh that follows; anything else in between drops it: driver.hip)
    } else
      LAUNCH(S, KC_MARKER_ADVECT, (k_advect_markers_b<true, false>), dim3(nb_b), dim3(256), in, out, S->uT, S->vT, S->solidT, gt, n, S->act_idx, S->act_dt, S->ms, (const unsigned int*)nullptr,
             (const uint8_t*)nullptr, (unsigned int*)nullptr, (unsigned long long*)nullptr, 0, (uint8_t*)nullptr, 0);
    S->cur ^= 1;
    return EULER_OK;
  }
  LAUNCH(S, KC_MARKER_ADVECT, k_advect_markers_a<false>, dim3(nb), dim3(256), in, out, S->u, S->v, S->solid, g, dt, n,
         S->evmask, S->ev_theta, S->ev_delta, S->ms);
  int rc = eu_ordered_select(S, S->evmask, (size_t)((n + 63) / 64), S->sel_idx, &S->ms->n_events);
  if (rc) return rc;
  LAUNCH(S, KC_MARKER_EVENTS, k_marker_walk, dim3(1), dim3(64), S->sel_idx, S->ev_theta, S->ev_delta, S->act_idx,
         S->act_dt, S->ms, dt);
  LAUNCH(S, KC_MARKER_ADVECT, (k_advect_markers_b<false, false>), dim3(nb_b), dim3(256), in, out, S->u, S->v, S->solid, g, n,
         S->act_idx, S->act_dt, S->ms, (const unsigned int*)nullptr, (const uint8_t*)nullptr, (unsigned int*)nullptr, (unsigned long long*)nullptr, 0, (uint8_t*)nullptr, 0);
  S->cur ^= 1;
  return EULER_OK;
}

// ==========================================================================================
// refresh_marker_counts (main.c:102-117)
//
// The 32-bit counters the markers are binned into are laid out COLUMN-major, [x][y - y0]: the marker array walks the cells
// column by column (seeding order, main.c:243-266, and the flow keeps neighbours in the array neighbours in space), so the
// 16 runs of a wave add into ONE 64-byte line instead of 16 lines X cells apart.  The adds are executed at the memory side,
// one transaction per wave instruction and line: 1.23 -> 0.14 ms for the 33 M adds of the 8192^2 half tank in isolation
// (tools/micro/atomic_bench.hip), k_bin_markers 3.4 -> 1.4 ms.  k_narrow_counts transposes back through LDS.
__global__ __launch_bounds__(256) void k_rotate_counts(uint8_t* __restrict__ prev, const uint8_t* __restrict__ cur,
                                                       unsigned int* __restrict__ count32, size_t i0, size_t C) {   // cells [i0, C): the window
  for (size_t i = i0 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < C; i += (size_t)gridDim.x * blockDim.x) {
    prev[i] = cur[i];
    count32[i - i0] = 0u;
  }
}

// sink | solid per cell, COLUMN-major like count32 (static between scenario edits: rebuilt when euler_sim.blocked_dirty says so).  The marker array walks the grid column by
// column, so a wave's 64 markers sit in ~16 cells of one column: in the row-major grids that is 16-17 cache lines per gather and two gathers per marker (1.37 ms at 8192^2,
// bound by lines per gather like k_advect_markers_a: profiles/r04_marker_gather_experiment.md); here it is one gather out of one or two lines, next to the counters' own line.
__global__ __launch_bounds__(256) void k_blocked_transpose(const uint8_t* __restrict__ sink, const uint8_t* __restrict__ solid, uint8_t* __restrict__ blockedT, int X, int Y) {
  __shared__ uint8_t tile[64][65];
  const int xb = blockIdx.x * 64, yb = blockIdx.y * 64, l = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int k = w; k < 64; k += 4) {                       // row yb + k, columns xb + l
    const int x = xb + l, y = yb + k;
    tile[k][l] = (x < X && y < Y) ? (uint8_t)((sink[(size_t)y * X + x] | solid[(size_t)y * X + x]) != 0) : (uint8_t)0;
  }
  __syncthreads();
  for (int k = w; k < 64; k += 4) {                       // column xb + k, rows yb + l
    const int x = xb + k, y = yb + l;
    if (x < X && y < Y) blockedT[(size_t)x * Y + y] = tile[l][k];
  }
}

__global__ __launch_bounds__(256) void k_bin_markers(const float2* __restrict__ m, unsigned long long n,
                                                     const uint8_t* __restrict__ blockedT,
                                                     unsigned int* count32, unsigned long long* __restrict__ delmask, int X, int H) {
  const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  bool del = false, live = false;
  size_t ct = 0;
  if (i < n) {
    const float2 p = m[i];
    const int x = (int)floorf(p.x / EU_H), y = (int)floorf(p.y / EU_H);
    ct = (size_t)x * H + y;
    del = blockedT[ct] != 0;      // refresh_marker_counts drops markers in sink or solid cells (main.c:109-112)
    live = !del;
  }
  bin_aggregated(count32, live, ct);
  const unsigned long long b = __ballot(del);
  if ((threadIdx.x & 63) == 0 && (i >> 6) < ((n + 63) >> 6)) delmask[i >> 6] = b;
}

// Swap-with-last deletion (main.c:112), in parallel.  The sequential loop leaves survivors with
// index < n' = n - D in place and fills the k-th hole (ascending) with the k-th survivor taken
// from the back (descending).  del_idx is ascending, so the holes are its first entries and a
// back survivor j finds its rank from the number of deletions above it.
__global__ __launch_bounds__(256) void k_compact_markers(float2* m, const unsigned int* __restrict__ del_idx,
                                                         const unsigned long long* __restrict__ delmask,
                                                         const MarkerState* ms) {
  const unsigned long long n = ms->n, D = ms->n_deleted;
  if (D == 0) return;
  const unsigned long long n1 = n - D;
  for (unsigned long long j = n1 + (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; j < n;
       j += (unsigned long long)gridDim.x * blockDim.x) {
    if ((delmask[j >> 6] >> (j & 63)) & 1ull) continue;   // deleted: dropped
    // deletions with index <= j
    unsigned long long lo = 0, hi = D;
    while (lo < hi) { const unsigned long long mid = (lo + hi) >> 1; if (del_idx[mid] <= j) lo = mid + 1; else hi = mid; }
    const unsigned long long del_above = D - lo;
    const unsigned long long rank = (n - 1 - j) - del_above;   // survivors behind j
    m[del_idx[rank]] = m[j];
  }
}

// count = (uint8_t)count32, transposed back to row-major through LDS: a workgroup takes 64 columns x 64 rows; it reads 64
// consecutive rows of a column with one wave (256 bytes) and writes 64 consecutive cells of a row with one wave (64 bytes).
// FOLD (whole-grid handles, round 6): the pass also does what k_rotate_counts did in front of the binning - prev <- cur (main.c:103) from the count grid it is
// about to overwrite, and cur <- 0 (main.c:104) for the counters it has just read: they are all zero again when the next refresh (or the advection pass that
// bins for it) starts.  One pass over the cells instead of two.
template <bool FOLD>
__global__ __launch_bounds__(256) void k_narrow_counts(uint8_t* __restrict__ count, unsigned int* __restrict__ count32,
                                                       int X, int y0, int y1, MarkerState* ms, int slab, uint8_t* __restrict__ prev,
                                                       uint8_t* __restrict__ tmap, int tnx, int tn, int known, int known_touch) {
  __shared__ uint8_t tile[64][65];
  const int bx = (int)blockIdx.x, by = (int)blockIdx.y;      // (tiles along a diagonal - the column-major side's power-of-two strides - measured: no difference)
  const int H = y1 - y0, xb = bx * 64, yb = by * 64;
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  bool nz = false;
  // (FOLD, the tile map: no water in the tile at the last refresh and no marker has entered it since - its counters are the zeros the last refresh left)
  const bool dry = FOLD && known && known_touch && tmap[by * tnx + bx] == 0 && tmap[2 * tn + by * tnx + bx] == 0;
  if (!dry)
    for (int k = w; k < 64; k += 4) {                     // column xb + k, rows yb + l
      const int x = xb + k, yr = yb + l;
      const bool in = x < X && yr < H;
      const unsigned int c = in ? count32[(size_t)x * H + yr] : 0u;
      tile[k][l] = (uint8_t)c;   // g_marker_count is uint8_t and wraps (main.c:96,114)
      if (FOLD && c != 0u) { count32[(size_t)x * H + yr] = 0u; nz = true; }
    }
  if (!FOLD) {
    __syncthreads();
    for (int k = w; k < 64; k += 4) {                     // row yb + k, columns xb + l
      const int x = xb + l, yr = yb + k;
      if (x < X && yr < H) count[(size_t)(y0 + yr) * X + x] = tile[l][k];
    }
  } else {
    // the tile map (euler_dev.h; the workgroup's tile IS a tile of the map: y0 = 0 on whole-grid handles): `known` - the flags describe the two grids as they stand;
    // a tile that held no water in either and receives none has nothing to move (its count and prev_count cells are the zeros they would get)
    const int t = by * tnx + bx;
    const int had = known ? (int)tmap[t] : 1, had_prev = known ? (int)tmap[tn + t] : 1;
    const int now = __syncthreads_or(nz ? 1 : 0);         // (a counter that wrapped to a multiple of 256 still sets the flag: a superset)
    bool oz = false;
    if ((had | had_prev | now) && (X & 3) == 0) {
      // four cells of a row per thread, as one 32-bit word (the byte-per-lane form below writes 64 bytes per wave instruction: 16384^2 dam break 443 -> 340 us, 8192^2 half tank 140 -> 106)
      const int xq = 4 * (int)(threadIdx.x & 15), r = (int)(threadIdx.x >> 4);
      for (int k = r; k < 64; k += 16) {                  // row yb + k, columns xb + xq .. xb + xq + 3
        const int x = xb + xq, yr = yb + k;
        if (x < X && yr < H) {
          const size_t i = (size_t)(y0 + yr) * X + x;
          const unsigned int oc = had ? *reinterpret_cast<const unsigned int*>(count + i) : 0u;
          oz = oz || oc != 0u;
          if (had | had_prev) *reinterpret_cast<unsigned int*>(prev + i) = oc;
          if (had | now)
            *reinterpret_cast<unsigned int*>(count + i) = (unsigned int)tile[xq][k] | ((unsigned int)tile[xq + 1][k] << 8) | ((unsigned int)tile[xq + 2][k] << 16) | ((unsigned int)tile[xq + 3][k] << 24);
        }
      }
    } else if (had | had_prev | now)
      for (int k = w; k < 64; k += 4) {                   // row yb + k, columns xb + l
        const int x = xb + l, yr = yb + k;
        if (x < X && yr < H) {
          const size_t i = (size_t)(y0 + yr) * X + x;
          const uint8_t oc = had ? count[i] : (uint8_t)0;
          oz = oz || oc != 0;
          if (had | had_prev) prev[i] = oc;
          if (had | now) count[i] = tile[l][k];
        }
      }
    const int was = known ? had : __syncthreads_or(oz ? 1 : 0);      // (`known` is uniform over the launch)
    if (threadIdx.x == 0) { tmap[tn + t] = (uint8_t)(was != 0); tmap[t] = (uint8_t)(now != 0); tmap[2 * tn + t] = 0; }
  }
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
    if (slab) { ms->n -= ms->n_del_glob; ms->n_loc -= ms->n_rm; }
    else ms->n -= ms->n_deleted;
  }
}

// pieces the row-slab substep (k_slab.hip) shares with the single-GPU stages: all over this rank's window of rows
int eu_marker_rotate_counts(euler_sim* S) {
  LAUNCH(S, KC_MARKER_BIN, k_rotate_counts, dim3(eu_blocks(S->Cw, 256 * 4, 4096)), dim3(256), S->prev_count, S->count,
         S->count32, S->win_off, (size_t)S->win_hi * S->X);
  return EULER_OK;
}
int eu_marker_narrow_counts(euler_sim* S) {
  LAUNCH(S, KC_MARKER_BIN, k_narrow_counts<false>, dim3((S->X + 63) / 64, (S->win_hi - S->win_lo + 63) / 64), dim3(256), S->count, S->count32, S->X,
         S->win_lo, S->win_hi, S->ms, S->slab_on, (uint8_t*)nullptr, (uint8_t*)nullptr, 0, 0, 0, 0);
  return EULER_OK;
}
int eu_marker_advect_a(euler_sim* S, float dt, unsigned long long n) {
  GridRef g{S->X, S->Y, S->count, S->interp_lim[0], S->interp_lim[1], S->interp_lim[2], S->interp_lim[3]};
  LAUNCH(S, KC_MARKER_ADVECT, k_advect_markers_a<false>, dim3(eu_blocks((size_t)n, 256)), dim3(256), S->markers[S->cur], S->markers[S->cur ^ 1], S->u, S->v, S->solid, g, dt, n,
         S->evmask, S->ev_theta, S->ev_delta, S->ms);
  return eu_ordered_select(S, S->evmask, (size_t)((n + 63) / 64), S->sel_idx, &S->ms->n_events);
}
int eu_marker_advect_b(euler_sim* S, unsigned long long n, const unsigned int* keys) {
  GridRef g{S->X, S->Y, S->count, S->interp_lim[0], S->interp_lim[1], S->interp_lim[2], S->interp_lim[3]};
  LAUNCH(S, KC_MARKER_ADVECT, (k_advect_markers_b<false, false>), dim3(eu_blocks((size_t)n, 256, 4096)), dim3(256), S->markers[S->cur], S->markers[S->cur ^ 1], S->u, S->v, S->solid, g, n,
         S->act_idx, S->act_dt, S->ms, keys, (const uint8_t*)nullptr, (unsigned int*)nullptr, (unsigned long long*)nullptr, 0, (uint8_t*)nullptr, 0);
  return EULER_OK;
}

int eu_launch_refresh_counts(euler_sim* S) {      // (whole-grid handles; row slabs: k_slab.hip)
  const unsigned long long n = S->n_markers_host;
  unsigned long long* delmask = S->evmask;
  int rc;
  const int touch_known = S->prebin_valid && S->opt[EULER_OPT_NO_TILE_MAP] == 0;      // (k_advect_bin_a2 / pass B marked every tile a marker entered)
  if (S->prebin_valid) delmask = S->delmask;      // the advection pass in front binned what it wrote (k_advect_bin_a2): the counters and the delete ballot stand
  else {
    if ((rc = eu_blocked_current(S)) || (rc = eu_count32_clean(S))) return rc;
    LAUNCH(S, KC_MARKER_BIN, k_bin_markers, dim3(eu_blocks((size_t)n, 256)), dim3(256), S->markers[S->cur], n, S->blockedT,
           S->count32, S->evmask, S->X, S->Y);
  }
  S->prebin_valid = 0;
  rc = eu_ordered_select(S, delmask, (size_t)((n + 63) / 64), S->sel_idx, &S->ms->n_deleted);
  if (rc) return rc;
  LAUNCH(S, KC_MARKER_COMPACT, k_compact_markers, dim3(256), dim3(256), S->markers[S->cur], S->sel_idx, delmask, S->ms);
  // prev <- cur, cur <- the counters, the counters <- 0
  LAUNCH(S, KC_MARKER_BIN, k_narrow_counts<true>, dim3((S->X + 63) / 64, (S->win_hi - S->win_lo + 63) / 64), dim3(256), S->count, S->count32, S->X,
         S->win_lo, S->win_hi, S->ms, 0, S->prev_count, S->tmap, S->tmap_nx, S->tmap_n, S->tmap_valid, touch_known);
  S->tmap_valid = 1;      // (both maps are what this pass saw; euler_set_option(EULER_OPT_NO_TILE_MAP) only stops the READERS)
  S->count32_dirty = 0;
  return EULER_OK;
}

// ==========================================================================================
// update_fluid_sources (main.c:276-298).  Eligible cells (source && count < 4) in row-major
// order each append one marker until the array holds MAX-1 markers (the latch, main.c:281,290).
// The k-th appended marker consumes draws 2k (y) and 2k+1 (x) of the xorshift64* stream: the
// compiled reference evaluates v2f(x+randf(), y+randf()) right to left.
__global__ __launch_bounds__(256) void k_source_mask(const uint8_t* __restrict__ source, const uint8_t* __restrict__ count,
                                                     size_t C, unsigned long long* __restrict__ mask) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool e = i < C && source[i] && count[i] < 4;
  const unsigned long long b = __ballot(e);
  if ((threadIdx.x & 63) == 0 && (i >> 6) < ((C + 63) >> 6)) mask[i >> 6] = b;
}

// the substep's bookkeeping (one thread): how many cells append (the latch, main.c:281,290), where the stream stands afterwards
__global__ void k_source_draws(MarkerState* ms, const RngJump* __restrict__ J) {
  unsigned long long n = ms->n;
  const unsigned long long cap = ms->max_markers - 1;
  int exhausted = ms->exhausted | (n == cap);
  unsigned long long n_app = 0;
  if (!exhausted) {
    n_app = ms->n_events;   // number of eligible cells (select total lands here)
    if (n_app > cap - n) n_app = cap - n;
  }
  ms->rng0 = ms->rng_state;
  ms->rng_state = eu_rng_jump(J, ms->rng_state, 2 * n_app);      // two draws per appended marker
  ms->n0_append = n;
  ms->n_append = (unsigned int)n_app;
  ms->src_k_lo = 0;
  n += n_app;
  ms->n = n;
  ms->exhausted = exhausted | (n == cap && n_app > 0) | (n == cap);
}
// ... and the draws themselves, in parallel: a thread jumps to its chunk of the ONE sequential stream (draw d is the generator's
// output after d + 1 steps from rng0) and walks SRC_CHUNK cells = 2 SRC_CHUNK draws.  Round 1 walked the stream with one
// thread: 6.9 ms per substep at 4096^2 waterfall (0.27 M source cells), 39 % of the run.
#define SRC_CHUNK 32
__global__ __launch_bounds__(256) void k_source_fill(const MarkerState* ms, const RngJump* __restrict__ J, float* __restrict__ draws) {
  const unsigned long long n_mine = ms->n_append, k_lo = ms->src_k_lo;        // this rank's cells are k_lo .. k_lo + n_mine of the substep's order
  const unsigned long long c0 = ((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x) * SRC_CHUNK;
  if (c0 >= n_mine) return;
  unsigned long long st = eu_rng_jump(J, ms->rng0, 2 * (k_lo + c0));
  const unsigned long long c1 = c0 + SRC_CHUNK < n_mine ? c0 + SRC_CHUNK : n_mine;
  for (unsigned long long c = c0; c < c1; ++c) {
    st = eu_rng_step(st); draws[2 * c] = eu_rng_float(st);          // y first (main.c:288: right to left)
    st = eu_rng_step(st); draws[2 * c + 1] = eu_rng_float(st);
  }
}

__global__ __launch_bounds__(256) void k_source_place(float2* __restrict__ m, uint8_t* __restrict__ count,
                                                      const unsigned int* __restrict__ elig, const float* __restrict__ draws,
                                                      const MarkerState* ms, int X, uint8_t* __restrict__ tmap, int tnx) {
  const unsigned int n_app = ms->n_append;
  const unsigned long long n0 = ms->n0_append;
  for (unsigned int k = blockIdx.x * blockDim.x + threadIdx.x; k < n_app; k += gridDim.x * blockDim.x) {
    const unsigned int c = elig[k];
    const int x = (int)(c % (unsigned)X), y = (int)(c / (unsigned)X);
    const float ry = draws[2 * k], rx = draws[2 * k + 1];
    m[n0 + k] = make_float2(EU_H * (x + rx), EU_H * (y + ry));
    count[c] = (uint8_t)(count[c] + 1);
    if (tmap) tmap[(y >> 6) * tnx + (x >> 6)] = 1;      // the tile map (euler_dev.h): the tile holds water now
  }
}

int eu_source_fill(euler_sim* S) {   // the parallel draws of this rank's appended markers (k_slab.hip shares it)
  LAUNCH(S, KC_SOURCES, k_source_fill, dim3(eu_blocks((S->n_source_cells + SRC_CHUNK - 1) / SRC_CHUNK, 256)), dim3(256), S->ms, S->rng_jump, S->draws);
  return EULER_OK;
}

int eu_launch_sources(euler_sim* S) {
  if (S->n_source_cells == 0) return EULER_OK;   // no '?' cells: the reference's loop body never runs
  LAUNCH(S, KC_SOURCES, k_source_mask, dim3(eu_blocks(S->C, 256)), dim3(256), S->source, S->count, S->C, S->cellmask64);
  int rc = eu_ordered_select(S, S->cellmask64, (S->C + 63) / 64, S->sel_idx, &S->ms->n_events);
  if (rc) return rc;
  LAUNCH(S, KC_SOURCES, k_source_draws, dim3(1), dim3(1), S->ms, S->rng_jump);
  eu_source_fill(S);
  LAUNCH(S, KC_SOURCES, k_source_place, dim3(eu_blocks(S->n_source_cells, 256, 2048)), dim3(256), S->markers[S->cur],
         S->count, S->sel_idx, S->draws, S->ms, S->X, S->slab_on ? (uint8_t*)nullptr : S->tmap, S->tmap_nx);
  return EULER_OK;
}
