// The grid-barrier question of scripts/microbench/grid_barrier.hip, asked again with the barrier the guide prescribes (VERDICT r5
// weak #10 / item 5): round 5 measured "+220 us per tick for a persistent probe + merge" with ONE counter that 2 048 blocks polled
// with atomicAdd(bar, 0) behind two __threadfence() each -- MI355X_MICROARCH.md's `barrier-counter` row in its worst form.  Here:
//   counter : one monotonic counter; lane 0: release fence (agent), relaxed atomic arrive, poll with a RELAXED sc1 LOAD + s_sleep,
//             acquire fence (agent)                                                             (the guide's barrier-counter row)
//   xcd     : XCD-hierarchical (the guide's barrier-xcd row): arrive at a per-XCC counter; the LAST arriver of an XCC is its
//             leader: release fence (one L2 write-back per XCD, not per block), arrive at the top counter, poll it, then publish
//             the XCC's generation word; everybody else polls its XCC's generation with relaxed loads; every block ends with an
//             acquire fence.  8 pollers on the top word, the rest spread over 8 lines.
// Same tick-shaped work as grid_barrier.hip (phase 1: P 64-bit atomicOr pushes to random members + one own word; phase 2: own
// words + P random cells of a 2 GB table read, one stored back; phase 2 must SEE phase 1 of the same tick: checked), as
//   two_kernels | one launch per tick with ONE barrier | one launch for all ticks with TWO barriers per tick,
// for grids of 1 / 2 / 4 workgroups per CU and 256- / 1024-thread workgroups, plus the bare barriers (empty phases).
// build: hipcc --offload-arch=gfx950 -O3 -o grid_barrier_xcd grid_barrier_xcd.hip     usage: grid_barrier_xcd [members_log2] [ticks]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int P = 3;
constexpr uint32_t LINE = 32;                     // words between two barrier words: each on a 128-byte line of its own
struct St { unsigned long long* inmask; unsigned long long* own; uint2* table; unsigned long long* out; uint32_t n; uint64_t tmask;
            unsigned* bar;                        // [0]: top counter; [(1 + x) * LINE]: XCC x's arrivals; [(9 + x) * LINE]: XCC x's generation; [17 * LINE ..]: census
            unsigned long long* bad; };
__device__ inline uint32_t mix32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ inline uint32_t xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u; }

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
template <int BLK> __global__ __launch_bounds__(BLK) void k_phase1(St s, uint32_t t) { phase1(s, blockIdx.x * BLK + threadIdx.x, t); }
template <int BLK> __global__ __launch_bounds__(BLK) void k_phase2(St s, uint32_t t) { phase2(s, blockIdx.x * BLK + threadIdx.x, t); }

__device__ inline unsigned ld_relaxed(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void spin_until(const unsigned* p, unsigned target) {
  // (signed distance: the words are monotonic and may wrap)
  for (uint32_t k = 0; (int)(ld_relaxed(p) - target) < 0; ++k) { __builtin_amdgcn_s_sleep(1); if (k > (1u << 26)) __builtin_trap(); }   // bounded: a stranded block must not hang the box
}
// gen = how many barriers this grid has passed, this one included (monotonic over the whole process: the words are never reset)
__device__ inline void barrier_counter(unsigned* bar, unsigned gen, unsigned nblocks) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    spin_until(bar, gen * nblocks);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}
__device__ inline void barrier_xcd(unsigned* bar, unsigned gen, unsigned per_xcc, unsigned n_xcc) {
  __syncthreads();                                  // (every wave's stores are acknowledged by the XCD's L2 behind this)
  if (threadIdx.x == 0) {
    const uint32_t x = xcc_id();
    const unsigned prev = __hip_atomic_fetch_add(&bar[(1u + x) * LINE], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (prev + 1u == gen * per_xcc) {               // the XCC's last arriver leads: ONE write-back of the XCD's L2
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __hip_atomic_fetch_add(&bar[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      spin_until(&bar[0], gen * n_xcc);
      __hip_atomic_store(&bar[(9u + x) * LINE], gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      spin_until(&bar[(9u + x) * LINE], gen);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}
// KIND 0: counter, 1: xcd.  WORK 0: empty phases (the bare barrier).
template <int BLK, int KIND, int WORK>
__global__ __launch_bounds__(BLK) void k_persistent(St s, uint32_t t0, uint32_t ticks, uint32_t members_tiles, unsigned gen0, unsigned per_xcc) {
  unsigned gen = gen0;
  auto bar = [&]() { ++gen; if (KIND == 0) barrier_counter(s.bar, gen, gridDim.x); else barrier_xcd(s.bar, gen, per_xcc, 8u); };
  for (uint32_t t = t0; t < t0 + ticks; ++t) {
    if (WORK) for (uint32_t b = blockIdx.x; b < members_tiles; b += gridDim.x) phase1(s, b * BLK + threadIdx.x, t);
    bar();
    if (WORK) for (uint32_t b = blockIdx.x; b < members_tiles; b += gridDim.x) phase2(s, b * BLK + threadIdx.x, t);
    if (t + 1 < t0 + ticks) bar();                 // the next tick's pushes must not overtake this tick's clears
  }
}
__global__ void k_census(unsigned* cen) { if (threadIdx.x == 0) atomicAdd(&cen[xcc_id()], 1u); }

template <int BLK, int KIND>
static void run(St s, const char* kind, int wg_per_cu, int cus, uint32_t T, uint32_t& t, unsigned* gens /* [2]: per kind */) {
  const uint32_t grid = (uint32_t)wg_per_cu * (uint32_t)cus, tiles = (s.n + BLK - 1) / BLK;
  int occ = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_persistent<BLK, KIND, 1>, BLK, 0));
  if (occ < wg_per_cu) { printf("# %s, %d-thread workgroups: %d per CU asked, occupancy %d: skipped\n", kind, BLK, wg_per_cu, occ); return; }
  // census: are the grid's blocks dealt evenly to the XCCs?  (the xcd barrier counts on grid / 8 arrivals per XCC)
  unsigned* cen = s.bar + 17 * LINE; unsigned h[8]; CK(hipMemset(cen, 0, 32));
  k_census<<<grid, BLK>>>(cen); CK(hipMemcpy(h, cen, 32, hipMemcpyDeviceToHost));
  bool even = true; for (int x = 0; x < 8; ++x) even = even && h[x] == grid / 8;
  if (!even && KIND == 1) { printf("# xcd barrier: blocks per XCC %u %u %u %u %u %u %u %u for a grid of %u: not even, skipped\n", h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], grid); return; }
  // the barrier words of both kinds restart for this grid size (gen * count must match this grid)
  CK(hipMemset(s.bar, 0, 17 * LINE * 4)); unsigned gen = 0; (void)gens;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); float ms;
  auto report = [&](const char* form, float us) { unsigned long long bad = 0; CK(hipMemcpy(&bad, s.bad, 8, hipMemcpyDeviceToHost));
    printf("{\"barrier\": \"%s\", \"threads_per_wg\": %d, \"wg_per_cu\": %d, \"form\": \"%s\", \"us_per_tick\": %.2f, \"phase2_saw_stale_phase1\": %llu}\n", kind, BLK, wg_per_cu, form, us, bad); fflush(stdout); };
  for (int rep = 0; rep < 2; ++rep) {
    // the bare barrier: 2T - 1 barriers, nothing between them
    CK(hipEventRecord(e0));
    k_persistent<BLK, KIND, 0><<<grid, BLK>>>(s, t, T, tiles, gen, grid / 8); gen += 2 * T - 1;
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    { unsigned long long bad = 0; (void)bad; printf("{\"barrier\": \"%s\", \"threads_per_wg\": %d, \"wg_per_cu\": %d, \"form\": \"bare barrier, empty phases\", \"us_per_barrier\": %.2f}\n", kind, BLK, wg_per_cu, ms * 1e3 / (2 * T - 1)); }
    CK(hipEventRecord(e0));
    for (uint32_t k = 0; k < T; ++k, ++t) { k_persistent<BLK, KIND, 1><<<grid, BLK>>>(s, t, 1u, tiles, gen, grid / 8); gen += 1; }
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); report("one launch per tick, 1 grid barrier", ms * 1e3 / T);
    CK(hipEventRecord(e0));
    k_persistent<BLK, KIND, 1><<<grid, BLK>>>(s, t, T, tiles, gen, grid / 8); gen += 2 * T - 1; t += T;
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); report("ONE launch for all ticks, 2 grid barriers per tick", ms * 1e3 / T);
  }
}

int main(int argc, char** argv) {
  const int lg = argc > 1 ? atoi(argv[1]) : 20; const uint32_t T = argc > 2 ? (uint32_t)atoi(argv[2]) : 50;
  St s{}; s.n = 1u << lg; const uint64_t cells = 1ull << 28; s.tmask = cells - 1;
  CK(hipMalloc(&s.inmask, (size_t)s.n * 8)); CK(hipMalloc(&s.own, (size_t)s.n * 8)); CK(hipMalloc(&s.out, (size_t)s.n * 8));
  CK(hipMalloc(&s.table, cells * 8)); CK(hipMalloc(&s.bar, 32 * LINE * 4)); CK(hipMalloc(&s.bad, 8));
  CK(hipMemset(s.inmask, 0, (size_t)s.n * 8)); CK(hipMemset(s.table, 0, cells * 8)); CK(hipMemset(s.bar, 0, 32 * LINE * 4)); CK(hipMemset(s.bad, 0, 8));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("# %u members, %u ticks per measurement, table 2^28 cells, %d CUs\n", s.n, T, cus);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); float ms;
  uint32_t t = 1; unsigned gens[2] = {0, 0};
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipEventRecord(e0));
    for (uint32_t k = 0; k < T; ++k, ++t) { k_phase1<256><<<(s.n + 255) / 256, 256>>>(s, t); k_phase2<256><<<(s.n + 255) / 256, 256>>>(s, t); }
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long bad = 0; CK(hipMemcpy(&bad, s.bad, 8, hipMemcpyDeviceToHost));
    printf("{\"form\": \"two_kernels\", \"threads_per_wg\": 256, \"us_per_tick\": %.2f, \"phase2_saw_stale_phase1\": %llu}\n", ms * 1e3 / T, bad);
  }
  for (int wg : {1, 2, 4}) { run<256, 0>(s, "counter", wg, cus, T, t, gens); run<256, 1>(s, "xcd", wg, cus, T, t, gens); }
  for (int wg : {1, 2}) { run<1024, 0>(s, "counter", wg, cus, T, t, gens); run<1024, 1>(s, "xcd", wg, cus, T, t, gens); }
  for (int wg : {8}) { run<256, 0>(s, "counter", wg, cus, T, t, gens); run<256, 1>(s, "xcd", wg, cus, T, t, gens); }
  return 0;
}
