// p2p_probe.hip — can two PROCESSES exchange flags through IPC-mapped device memory while both have a
// kernel resident (the mailbox all-reduce of csrc/comm_p2p.hip)?  Measures the ping-pong latency.
//   ./p2p_probe <rank 0|1> <dir> [peer_device]     (start both; they meet through files in <dir>)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <unistd.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e)); exit(2); } } while (0)

__global__ void k_pingpong(unsigned long long* mine, unsigned long long* peer, int rank, int rounds, unsigned long long* out) {
  unsigned long long t0 = wall_clock64();
  unsigned int fails = 0;
  for (int r = 1; r <= rounds; ++r) {
    if (rank == 0) {
      __hip_atomic_store(peer, (unsigned long long)r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      unsigned int spins = 0;
      while (__hip_atomic_load(mine, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < (unsigned long long)r) if (++spins > (1u << 26)) { fails++; break; }
    } else {
      unsigned int spins = 0;
      while (__hip_atomic_load(mine, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < (unsigned long long)r) if (++spins > (1u << 26)) { fails++; break; }
      __hip_atomic_store(peer, (unsigned long long)r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (fails) break;
  }
  out[0] = wall_clock64() - t0;
  out[1] = fails;
}

static bool wait_file(const std::string& p, void* buf, size_t n) {
  for (int k = 0; k < 3000; ++k) {
    FILE* f = fopen(p.c_str(), "rb");
    if (f) { size_t got = fread(buf, 1, n, f); fclose(f); if (got == n) return true; }
    std::this_thread::sleep_for(std::chrono::milliseconds(10));
  }
  return false;
}

int main(int argc, char** argv) {
  int rank = atoi(argv[1]);
  std::string dir = argv[2];
  int ndev = 0;
  CK(hipGetDeviceCount(&ndev));
  int dev = ndev > 1 ? rank : 0;
  CK(hipSetDevice(dev));
  for (int mode = 0; mode < 2; ++mode) {   // 0: fine-grained allocation, 1: plain hipMalloc
    unsigned long long* box = nullptr;
    if (mode == 0) { hipError_t e = hipExtMallocWithFlags((void**)&box, 4096, hipDeviceMallocFinegrained); if (e != hipSuccess) { printf("rank %d mode 0: finegrained alloc failed: %s\n", rank, hipGetErrorString(e)); continue; } }
    else CK(hipMalloc((void**)&box, 4096));
    CK(hipMemset(box, 0, 4096));
    hipIpcMemHandle_t h;
    hipError_t e = hipIpcGetMemHandle(&h, box);
    if (e != hipSuccess) { printf("rank %d mode %d: hipIpcGetMemHandle: %s\n", rank, mode, hipGetErrorString(e)); continue; }
    std::string mine = dir + "/h" + std::to_string(mode) + "_" + std::to_string(rank), theirs = dir + "/h" + std::to_string(mode) + "_" + std::to_string(1 - rank);
    { std::string tmp = mine + ".tmp"; FILE* f = fopen(tmp.c_str(), "wb"); fwrite(&h, sizeof h, 1, f); fclose(f); rename(tmp.c_str(), mine.c_str()); }
    hipIpcMemHandle_t ph;
    if (!wait_file(theirs, &ph, sizeof ph)) { printf("rank %d: peer handle never came\n", rank); return 3; }
    unsigned long long* peer = nullptr;
    e = hipIpcOpenMemHandle((void**)&peer, ph, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) { printf("rank %d mode %d: hipIpcOpenMemHandle: %s\n", rank, mode, hipGetErrorString(e)); continue; }
    unsigned long long* out;
    CK(hipMalloc((void**)&out, 16));
    const int rounds = 2000;
    hipLaunchKernelGGL(k_pingpong, dim3(1), dim3(1), 0, 0, box, peer, rank, rounds, out);
    CK(hipDeviceSynchronize());
    unsigned long long res[2];
    CK(hipMemcpy(res, out, 16, hipMemcpyDeviceToHost));
    printf("rank %d mode %s: %d round trips in %.1f us -> %.2f us per round trip, timeouts %llu\n", rank, mode == 0 ? "finegrained" : "hipMalloc",
           rounds, res[0] / 100.0, res[0] / 100.0 / rounds, res[1]);
    fflush(stdout);
    CK(hipIpcCloseMemHandle(peer));
    // keep `box` alive until the peer is certainly done
    std::string done = dir + "/d" + std::to_string(mode) + "_" + std::to_string(rank), pdone = dir + "/d" + std::to_string(mode) + "_" + std::to_string(1 - rank);
    { FILE* f = fopen(done.c_str(), "wb"); fputc(1, f); fclose(f); }
    char c; wait_file(pdone, &c, 1);
  }
  return 0;
}
