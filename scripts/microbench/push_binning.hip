// What would taking probe_kernel's Ping pushes off the global-atomic rate buy?    The probe delivers a Ping's
// piggyback payload as ONE 64-bit atomicOr into inmask[target]: ~3.1 M scattered atomics per launch at a million members, and the
// chip serves scattered atomics at a fixed ~27 G/s whatever the table size or scope (profiles/r01_microbench_*).  The alternative:
// the pushing workgroup BINS its {target, mask} records by destination region in LDS, reserves room per (workgroup, region) with
// one atomic, stores the records in runs; a CONSUMER kernel ORs a region's records into an LDS tile of inmask with LDS atomics and
// writes the tile out.  This file has both forms without the rest of the probe, same pushes, results compared word by word:
//   atomics      : one thread per member, P atomicOr(inmask[dst], mask)
//   binned R     : binning kernel (R regions of N / R members) + consumer kernel (one workgroup per region, the region's inmask in LDS)
// build: hipcc --offload-arch=gfx950 -O3 -o push_binning push_binning.hip      usage: push_binning [members_log2] [rounds]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int P = 3, BLK = 256;
__device__ inline uint32_t mix32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ inline void push_of(uint32_t i, int p, uint32_t t, uint32_t n, uint32_t* dst, unsigned long long* mask) {
  *dst = (uint32_t)(((uint64_t)mix32(i * 3u + p + t * 0x9E3779B9u) * n) >> 32);
  *mask = (1ull << ((i + t) & 63u)) | (1ull << ((i * 7u + p) & 63u));
}

__global__ __launch_bounds__(BLK) void k_atomics(unsigned long long* inmask, uint32_t n, uint32_t t) {
  const uint32_t i = blockIdx.x * BLK + threadIdx.x;
  if (i >= n) return;
  for (int p = 0; p < P; ++p) { uint32_t d; unsigned long long m; push_of(i, p, t, n, &d, &m); atomicOr(&inmask[d], m); }
}

// binning: the workgroup's BLK * P records, counted per region in LDS, one reservation per (workgroup, region), stored in runs
struct Rec { uint32_t dst, pad; unsigned long long mask; };
template <int R>
__global__ __launch_bounds__(BLK) void k_bin(Rec* recs, uint32_t* rcount, uint32_t cap, uint32_t n, uint32_t t, uint32_t region_shift) {
  __shared__ uint32_t cnt[R], base[R];
  for (uint32_t r = threadIdx.x; r < R; r += BLK) cnt[r] = 0;
  __syncthreads();
  const uint32_t i = blockIdx.x * BLK + threadIdx.x;
  uint32_t d[P], off[P]; unsigned long long m[P];
  for (int p = 0; p < P; ++p) {
    d[p] = 0; m[p] = 0; off[p] = 0;
    if (i < n) { push_of(i, p, t, n, &d[p], &m[p]); off[p] = atomicAdd(&cnt[d[p] >> region_shift], 1u); }
  }
  __syncthreads();
  for (uint32_t r = threadIdx.x; r < R; r += BLK) base[r] = cnt[r] ? atomicAdd(&rcount[r * 32u], cnt[r]) : 0u;   // (counters on lines of their own)
  __syncthreads();
  if (i < n)
    for (int p = 0; p < P; ++p) {
      const uint32_t r = d[p] >> region_shift, at = base[r] + off[p];
      if (at < cap) recs[(size_t)r * cap + at] = Rec{d[p], 0u, m[p]};
    }
}
// consumer: one workgroup per region, the region's slice of inmask in LDS (dynamic), OR-ed with LDS atomics, written out coalesced
template <int CBLK>
__global__ __launch_bounds__(CBLK) void k_consume(const Rec* recs, uint32_t* rcount, uint32_t cap, unsigned long long* inmask, uint32_t region_members) {
  extern __shared__ unsigned long long tile[];
  const uint32_t r = blockIdx.x;
  for (uint32_t k = threadIdx.x; k < region_members; k += CBLK) tile[k] = inmask[(size_t)r * region_members + k];
  __syncthreads();
  const uint32_t nrec = min(rcount[r * 32u], cap);
  const Rec* mine = recs + (size_t)r * cap;
  for (uint32_t k = threadIdx.x; k < nrec; k += CBLK) { const Rec e = mine[k]; atomicOr(&tile[e.dst & (region_members - 1u)], e.mask); }
  __syncthreads();
  for (uint32_t k = threadIdx.x; k < region_members; k += CBLK) inmask[(size_t)r * region_members + k] = tile[k];
  if (threadIdx.x == 0) rcount[r * 32u] = 0;
}

template <int R, int CBLK>
static float run_binned(Rec* recs, uint32_t* rcount, uint32_t cap, unsigned long long* inmask, uint32_t n, uint32_t t0, int rounds, hipEvent_t e0, hipEvent_t e1) {
  const uint32_t region_members = n / R; uint32_t shift = 0; while ((1u << shift) < region_members) ++shift;
  const size_t lds = (size_t)region_members * 8;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_consume<CBLK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  float ms;
  CK(hipEventRecord(e0));
  for (int k = 0; k < rounds; ++k) {
    k_bin<R><<<(n + BLK - 1) / BLK, BLK>>>(recs, rcount, cap, n, t0 + k, shift);
    k_consume<CBLK><<<R, CBLK, lds>>>(recs, rcount, cap, inmask, region_members);
  }
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3f / rounds;
}

int main(int argc, char** argv) {
  const int lg = argc > 1 ? atoi(argv[1]) : 20; const int rounds = argc > 2 ? atoi(argv[2]) : 50;
  const uint32_t n = 1u << lg;
  unsigned long long *a, *b; Rec* recs; uint32_t* rcount;
  const uint32_t cap_total = n * P * 2;             // records, with slack per region
  CK(hipMalloc(&a, (size_t)n * 8)); CK(hipMalloc(&b, (size_t)n * 8)); CK(hipMalloc(&recs, (size_t)cap_total * sizeof(Rec))); CK(hipMalloc(&rcount, 1024 * 32 * 4));
  CK(hipMemset(rcount, 0, 1024 * 32 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); float ms;
  printf("# %u members, %d pushes each (every one a nonzero mask), %d rounds per measurement\n", n, P, rounds);
  std::vector<unsigned long long> ha(n), hb(n);
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipMemset(a, 0, (size_t)n * 8));
    CK(hipEventRecord(e0));
    for (int k = 0; k < rounds; ++k) k_atomics<<<(n + BLK - 1) / BLK, BLK>>>(a, n, 100 + k);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf("{\"form\": \"atomics\", \"us_per_launch\": %.2f, \"G_atomics_per_s\": %.1f}\n", ms * 1e3 / rounds, (double)n * P * rounds / (ms * 1e-3) / 1e9);
    CK(hipMemcpy(ha.data(), a, (size_t)n * 8, hipMemcpyDeviceToHost));
    auto check = [&](const char* name, float us) {
      CK(hipMemcpy(hb.data(), b, (size_t)n * 8, hipMemcpyDeviceToHost));
      size_t bad = 0; for (uint32_t k = 0; k < n; ++k) bad += ha[k] != hb[k];
      printf("{\"form\": \"%s\", \"us_per_tick_both_kernels\": %.2f, \"words_that_differ_from_the_atomics\": %zu}\n", name, us, bad); fflush(stdout);
    };
    CK(hipMemset(b, 0, (size_t)n * 8)); check("binned, 64 regions of 16384 members (128 KB LDS tile, 1024-thread consumers)", run_binned<64, 1024>(recs, rcount, cap_total / 64, b, n, 100, rounds, e0, e1));
    CK(hipMemset(b, 0, (size_t)n * 8)); check("binned, 128 regions of 8192 members (64 KB LDS tile, 1024-thread consumers)", run_binned<128, 1024>(recs, rcount, cap_total / 128, b, n, 100, rounds, e0, e1));
    CK(hipMemset(b, 0, (size_t)n * 8)); check("binned, 256 regions of 4096 members (32 KB LDS tile, 512-thread consumers)", run_binned<256, 512>(recs, rcount, cap_total / 256, b, n, 100, rounds, e0, e1));
  }
  // the kernels apart (64 regions)
  {
    const uint32_t R = 64, rm = n / R; uint32_t shift = 0; while ((1u << shift) < rm) ++shift;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_consume<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(rm * 8)));
    float tb = 0, tc = 0;
    for (int k = 0; k < rounds; ++k) {
      CK(hipEventRecord(e0)); k_bin<64><<<(n + BLK - 1) / BLK, BLK>>>(recs, rcount, cap_total / 64, n, 100 + k, shift); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); tb += ms;
      CK(hipEventRecord(e0)); k_consume<1024><<<R, 1024, rm * 8>>>(recs, rcount, cap_total / 64, b, rm); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); tc += ms;
    }
    printf("{\"form\": \"binned, 64 regions, kernels timed apart\", \"bin_us\": %.2f, \"consume_us\": %.2f}\n", tb * 1e3 / rounds, tc * 1e3 / rounds);
  }
  return 0;
}
