// How many random 8-byte accesses per second does an MI355X serve?  (what bounds merge_kernel.)
// Each lane draws addresses from a counter hash over a table of `cells` 8-byte cells (default 1 G cells = 8 GB: merge_kernel's view
// rows at a million members), B independent loads in flight per lane and round (B = 1, 2, 4), optionally stores the cell back
// changed (the state rule: load -> compare -> store).  Reported: accesses / s, the 64-byte sectors they touch per second.
// build: hipcc --offload-arch=gfx950 -O3 -o random_access random_access.hip      usage: random_access [cells_log2] [rounds]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

__device__ inline uint32_t mix32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int B, bool STORE, int GROUP>   // GROUP: lanes of a wave that share a 64-byte sector (1 = every lane its own; 8 = eight neighbours)
__global__ __launch_bounds__(256) void gather(uint2* tab, uint64_t mask, uint32_t rounds, uint32_t seed, unsigned long long* sink) {
  const uint32_t tid = blockIdx.x * 256u + threadIdx.x;
  unsigned long long acc = 0;
  for (uint32_t r = 0; r < rounds; ++r) {
    uint2 v[B]; uint64_t a[B];
#pragma unroll
    for (int k = 0; k < B; ++k) {
      const uint32_t g = tid / GROUP, l = tid % GROUP;
      const uint64_t h = ((uint64_t)mix32(g * 0x9E3779B1u + seed + r * B + k) << 20) ^ mix32(g + 77u * (r * B + k) + seed);
      a[k] = ((h * GROUP) + l) & mask;
      v[k] = tab[a[k]];
    }
#pragma unroll
    for (int k = 0; k < B; ++k) {
      acc += v[k].x;
      if (STORE && (v[k].x & 1u) == 0u) tab[a[k]] = make_uint2(v[k].x + 2u, r);
    }
  }
  if (acc == 0x123456789ull) *sink = acc;
}

template <int B, bool STORE, int GROUP>
static void run(const char* name, uint2* tab, uint64_t cells, uint32_t rounds, unsigned long long* sink) {
  const uint32_t blocks = 4096;   // 16 384 waves, as merge_kernel at a million members
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  gather<B, STORE, GROUP><<<blocks, 256>>>(tab, cells - 1, rounds, 1u, sink);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    gather<B, STORE, GROUP><<<blocks, 256>>>(tab, cells - 1, rounds, 100u + rep, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  const double acc = (double)blocks * 256.0 * rounds * B;
  printf("{\"case\": \"%s\", \"in_flight_per_lane\": %d, \"store_back\": %s, \"lanes_per_sector\": %d, \"us\": %.1f, \"G_accesses_per_s\": %.1f, \"G_sectors_per_s\": %.1f}\n",
         name, B, STORE ? "true" : "false", GROUP, best * 1e3, acc / best / 1e6, acc / GROUP / best / 1e6);
}

int main(int argc, char** argv) {
  const int lg = argc > 1 ? atoi(argv[1]) : 30;
  const uint32_t rounds = argc > 2 ? (uint32_t)atoi(argv[2]) : 8;
  const uint64_t cells = 1ull << lg;
  uint2* tab; unsigned long long* sink;
  if (hipMalloc(&tab, cells * 8) != hipSuccess) { fprintf(stderr, "alloc failed\n"); return 1; }
  hipMalloc(&sink, 8);
  hipMemset(tab, 0, cells * 8);
  printf("# table: 2^%d cells of 8 B = %.1f GB; 16 384 waves x %u rounds\n", lg, cells * 8 / 1e9, rounds);
  run<1, false, 1>("loads", tab, cells, rounds, sink);
  run<2, false, 1>("loads", tab, cells, rounds, sink);
  run<4, false, 1>("loads", tab, cells, rounds, sink);
  run<8, false, 1>("loads", tab, cells, rounds, sink);
  run<2, true, 1>("load + store back", tab, cells, rounds, sink);
  run<4, true, 1>("load + store back", tab, cells, rounds, sink);
  run<2, false, 8>("loads, 8 neighbours share a sector", tab, cells, rounds, sink);
  run<2, true, 8>("load + store back, 8 neighbours share a sector", tab, cells, rounds, sink);
  return 0;
}
