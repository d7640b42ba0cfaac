// poll_bench.hip — what does one hop of the band hand-off cost, by poll path and placement?
// Producer and consumer waves on different CUs play ping-pong through 16-byte granules {lo, tag, hi, tag}.
//   poll path:  0 = vector load sc1 (today's fetch wave), 1 = vector load sc0 (L2 hit when both sit on one XCD),
//               2 = scalar load glc (own queue: not behind the CU's vector-memory traffic)
//   store:      0 = sc1 write-through (today), 1 = plain
//   placement:  same XCD / neighbouring XCDs
//   noise:      a second wave on each CU streaming 16-byte loads from a 256 MB buffer (the compute wave's streams)
// Prints the one-way latency in ns (wall clock 100 MHz).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e)); exit(2); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct Ctl { unsigned int slots[8]; unsigned int stop; unsigned int pad[7]; unsigned long long t[16]; unsigned int fail[16]; };

template <int POLL>
__device__ __forceinline__ bool poll(const u32x4* p, unsigned int tag) {
  u32x4 g;
  if (POLL == 0) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(g) : "v"(p) : "memory");
  else if (POLL == 1) asm volatile("global_load_dwordx4 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(g) : "v"(p) : "memory");
  else {
    typedef unsigned int s4 __attribute__((ext_vector_type(4)));
    s4 sg;
    const unsigned long long pa = (unsigned long long)p;
    const unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)pa), hi = __builtin_amdgcn_readfirstlane((unsigned int)(pa >> 32));
    const unsigned long long sp = ((unsigned long long)hi << 32) | lo;
    asm volatile("s_load_dwordx4 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(sg) : "s"(sp) : "memory");
    g = sg;
  }
  return g[1] == tag && g[3] == tag;
}
template <int ST>
__device__ __forceinline__ void put(u32x4* p, unsigned int tag) {
  const u32x4 g = {tag * 3u, tag, tag * 7u, tag};
  if (ST == 0) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(g) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(p), "v"(g) : "memory");
}

template <int POLL, int ST>
__global__ __launch_bounds__(128) void k(Ctl* c, u32x4* box, const u32x4* big, size_t big_n, int rounds, int cross, int noise) {
  __shared__ int role;   // -1 idle, 0 producer, 1 consumer
  __shared__ int pair;
  if (threadIdx.x == 0) {
    unsigned int xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7;
    const unsigned int k = atomicAdd(&c->slots[xcc], 1u);
    role = k == 0 ? 0 : (k == 1 ? 1 : -1);
    // same-XCD: producer and consumer of XCD x talk; cross: consumer of XCD x answers the producer of XCD x-1
    pair = role == 0 ? (int)xcc : (cross ? (int)((xcc + 7) & 7) : (int)xcc);
  }
  __syncthreads();
  if (role < 0) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  u32x4* a = box + pair * 64;        // producer -> consumer   (separate 1 KB blocks)
  u32x4* b = box + 8 * 64 + pair * 64;   // consumer -> producer
  if (wave == 1) {                   // noise: keep 4 streaming loads in flight until told to stop
    if (!noise) return;
    size_t i = ((size_t)blockIdx.x * 977 + lane) % big_n;
    u32x4 acc = {0, 0, 0, 0};
    while (__hip_atomic_load(&c->stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 16u) {
      for (int r = 0; r < 8; ++r) { acc += big[i]; i = (i + 64 * 131) % big_n; }
    }
    if (acc[0] == 0x12345u) box[1000] = acc;
    return;
  }
  if (lane != 0) return;
  // wait until every XCD has both actors (so that the pairs exist), bounded
  for (int s = 0; s < (1 << 22); ++s) {
    bool all = true;
    for (int x = 0; x < 8; ++x) all = all && __hip_atomic_load(&c->slots[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= 2u;
    if (all) break;
  }
  unsigned int fails = 0;
  const unsigned long long t0 = wall_clock64();
  for (int r = 1; r <= rounds && !fails; ++r) {
    if (role == 0) {
      put<ST>(a, (unsigned int)r);
      unsigned int sp = 0;
      while (!poll<POLL>(b, (unsigned int)r)) if (++sp > (1u << 22)) { fails = r; break; }
    } else {
      unsigned int sp = 0;
      while (!poll<POLL>(a, (unsigned int)r)) if (++sp > (1u << 22)) { fails = r; break; }
      put<ST>(b, (unsigned int)r);
    }
  }
  const unsigned long long t1 = wall_clock64();
  if (role == 0) { c->t[pair] = t1 - t0; c->fail[pair] = fails; } else if (fails) c->fail[8 + pair] = fails;
  atomicAdd(&c->stop, 1u);
}

template <int POLL, int ST>
static void run(const char* name, Ctl* c, u32x4* box, u32x4* big, size_t big_n, int cross, int noise) {
  CK(hipMemset(c, 0, sizeof(Ctl)));
  CK(hipMemset(box, 0, 64 * 1024));
  const int rounds = 2000;
  hipLaunchKernelGGL((k<POLL, ST>), dim3(64), dim3(128), 0, 0, c, box, big, big_n, rounds, cross, noise);
  CK(hipDeviceSynchronize());
  Ctl h;
  CK(hipMemcpy(&h, c, sizeof h, hipMemcpyDeviceToHost));
  double best = 1e30, sum = 0; int n = 0, bad = 0;
  for (int x = 0; x < 8; ++x) { if (h.fail[x] || h.fail[8 + x]) { bad++; continue; } double ns = h.t[x] * 10.0 / (2.0 * rounds); sum += ns; n++; if (ns < best) best = ns; }
  printf("%-28s %-10s noise %d : one-way %7.0f ns mean, %7.0f ns best over %d pairs%s\n", name, cross ? "cross-XCD" : "same-XCD", noise, n ? sum / n : 0.0, n ? best : 0.0, n,
         bad ? "  (some pairs timed out: stale reads)" : "");
}

int main() {
  Ctl* c; u32x4 *box, *big;
  const size_t big_n = (256u << 20) / 16;
  CK(hipMalloc((void**)&c, sizeof(Ctl))); CK(hipMalloc((void**)&box, 64 * 1024)); CK(hipMalloc((void**)&big, big_n * 16));
  CK(hipMemset(big, 1, big_n * 16));
  for (int noise = 0; noise < 2; ++noise)
    for (int cross = 0; cross < 2; ++cross) {
      run<0, 0>("vector sc1 / store sc1", c, box, big, big_n, cross, noise);
      run<0, 1>("vector sc1 / store plain", c, box, big, big_n, cross, noise);
      run<1, 1>("vector sc0 / store plain", c, box, big, big_n, cross, noise);
      run<1, 0>("vector sc0 / store sc1", c, box, big, big_n, cross, noise);
      run<2, 1>("scalar glc / store plain", c, box, big, big_n, cross, noise);
      run<2, 0>("scalar glc / store sc1", c, box, big, big_n, cross, noise);
    }
  return 0;
}
