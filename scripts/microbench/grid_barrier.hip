// What would ONE launch per tick buy?    The tick's two big kernels are separated by a
// kernel boundary (~6 us on this chip: every XCD's L2 is written back and invalidated).  A persistent probe + merge kernel replaces
// that boundary by a GRID BARRIER, which has to do the same release / acquire itself.  This microbenchmark has the tick's shape
// without its logic: phase 1 = one thread per "member" pushes a 64-bit atomicOr to P random members' words and stores a word of
// its own (the probe's traffic), phase 2 = every member reads its own words and P random cells of a big table and stores a result
// (the merge's traffic); phase 2 must SEE phase 1 (checked).  Three forms, same work, same grid of 256-thread blocks:
//   two_kernels    : phase 1 and phase 2 as two launches on one stream (what the library does);
//   persistent     : one launch, grid = what the chip holds at once, every block walks its tiles of phase 1, a grid barrier
//                    (arrive: __threadfence + atomic add; wait: spin on the counter, then __threadfence), then its tiles of phase 2;
//   persistent x T : the same kernel looping over T "ticks" inside one launch (two barriers per tick): no launch at all.
// build: hipcc --offload-arch=gfx950 -O3 -o grid_barrier grid_barrier.hip        usage: grid_barrier [members_log2] [ticks]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int P = 3, BLK = 256;
struct St { unsigned long long* inmask; unsigned long long* own; uint2* table; unsigned long long* out; uint32_t n; uint64_t tmask; unsigned* bar; unsigned long long* bad; };
__device__ inline uint32_t mix32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

__device__ inline void phase1(const St& s, uint32_t i, uint32_t t) {
  if (i >= s.n) return;
  for (int p = 0; p < P; ++p) {
    const uint32_t j = (uint32_t)(((uint64_t)mix32(i * 3u + p + t * 0x9E3779B9u) * s.n) >> 32);
    atomicOr(&s.inmask[j], 1ull << ((i + t) & 63u));
  }
  s.own[i] = ((unsigned long long)t << 32) | i;
}
__device__ inline void phase2(const St& s, uint32_t i, uint32_t t) {
  if (i >= s.n) return;
  unsigned long long acc = s.inmask[i];
  const unsigned long long mine = s.own[i];
  if (mine != (((unsigned long long)t << 32) | i)) atomicAdd(s.bad, 1ull);     // phase 1 of THIS tick must be visible
  for (int p = 0; p < P; ++p) {
    const uint64_t a = (((uint64_t)mix32(i + 77u * p + t) << 20) ^ mix32(i * 5u + p)) & s.tmask;
    const uint2 c = s.table[a];
    acc += c.x;
    if ((c.x & 1u) == 0u) s.table[a] = make_uint2(c.x + 2u, t);
  }
  s.inmask[i] = 0;
  s.out[i] = acc;
}
__global__ __launch_bounds__(BLK) void k_phase1(St s, uint32_t t) { phase1(s, blockIdx.x * BLK + threadIdx.x, t); }
__global__ __launch_bounds__(BLK) void k_phase2(St s, uint32_t t) { phase2(s, blockIdx.x * BLK + threadIdx.x, t); }

__device__ inline void grid_barrier(unsigned* bar, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();                                   // release: this block's stores (and the XCD's dirty lines) become visible to the agent
    atomicAdd(bar, 1u);
    while (atomicAdd(bar, 0u) < target) __builtin_amdgcn_s_sleep(2);
    __threadfence();                                   // acquire
  }
  __syncthreads();
}
__global__ __launch_bounds__(BLK) void k_persistent(St s, uint32_t t0, uint32_t ticks, uint32_t tiles, unsigned base) {
  unsigned done = base;
  for (uint32_t t = t0; t < t0 + ticks; ++t) {
    for (uint32_t b = blockIdx.x; b < tiles; b += gridDim.x) phase1(s, b * BLK + threadIdx.x, t);
    done += gridDim.x; grid_barrier(s.bar, done);
    for (uint32_t b = blockIdx.x; b < tiles; b += gridDim.x) phase2(s, b * BLK + threadIdx.x, t);
    if (t + 1 < t0 + ticks) { done += gridDim.x; grid_barrier(s.bar, done); }      // the next tick's pushes must not overtake this tick's clears
  }
}

int main(int argc, char** argv) {
  const int lg = argc > 1 ? atoi(argv[1]) : 20; const uint32_t T = argc > 2 ? (uint32_t)atoi(argv[2]) : 50;
  St s{}; s.n = 1u << lg; const uint64_t cells = 1ull << 28; s.tmask = cells - 1;
  CK(hipMalloc(&s.inmask, (size_t)s.n * 8)); CK(hipMalloc(&s.own, (size_t)s.n * 8)); CK(hipMalloc(&s.out, (size_t)s.n * 8));
  CK(hipMalloc(&s.table, cells * 8)); CK(hipMalloc(&s.bar, 4)); CK(hipMalloc(&s.bad, 8));
  CK(hipMemset(s.inmask, 0, (size_t)s.n * 8)); CK(hipMemset(s.table, 0, cells * 8)); CK(hipMemset(s.bar, 0, 4)); CK(hipMemset(s.bad, 0, 8));
  const uint32_t tiles = (s.n + BLK - 1) / BLK;
  int perCU = 0, dev = 0; hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, dev));
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, k_persistent, BLK, 0));
  const uint32_t grid = (uint32_t)perCU * (uint32_t)prop.multiProcessorCount;
  printf("# %u members, %u tiles of %d; persistent grid = %d blocks/CU x %d CUs = %u blocks; %u ticks per measurement; table 2^28 cells\n", s.n, tiles, BLK, perCU, prop.multiProcessorCount, grid, T);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  uint32_t t = 1; unsigned base = 0; float ms;
  auto report = [&](const char* name, float ms_) { unsigned long long bad = 0; CK(hipMemcpy(&bad, s.bad, 8, hipMemcpyDeviceToHost));
    printf("{\"form\": \"%s\", \"us_per_tick\": %.2f, \"phase2_saw_stale_phase1\": %llu}\n", name, ms_ * 1e3 / T, bad); };
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    for (uint32_t k = 0; k < T; ++k, ++t) { k_phase1<<<tiles, BLK>>>(s, t); k_phase2<<<tiles, BLK>>>(s, t); }
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); report("two_kernels", ms);
    CK(hipEventRecord(e0));
    for (uint32_t k = 0; k < T; ++k, ++t) { k_persistent<<<grid, BLK>>>(s, t, 1u, tiles, base); base += grid; }
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); report("persistent, one launch per tick (1 grid barrier)", ms);
    CK(hipEventRecord(e0));
    k_persistent<<<grid, BLK>>>(s, t, T, tiles, base); base += grid * (2 * T - 1); t += T;
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); report("persistent, ONE launch for all ticks (2 grid barriers per tick)", ms);
  }
  return 0;
}
