// Microbenchmark (measurement tool, not product): can scattered atomics run in the XCD-local L2 instead
// of the memory side?  Each XCD gets its own copy of the table (selected by the hardware XCC_ID) and uses
// workgroup-scope atomics; a reader then combines the 8 copies.  Checks the result is exact.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__device__ inline uint32_t mix32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ inline uint32_t xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u; }
// mode 0: agent-scope 64-bit atomicOr+count on one table; 1: workgroup-scope on per-XCD copy; 2: agent-scope on per-XCD copy
template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned long long* tab, uint32_t n, uint32_t seed, uint32_t* xhist) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  const uint32_t x = xcc_id();
  if (threadIdx.x == 0) atomicAdd(&xhist[x * 16 + (blockIdx.x & 7u)], 1u);
  unsigned long long* base = MODE == 0 ? tab : tab + (size_t)x * n;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const uint32_t idx = __umulhi(mix32(mix32(i ^ seed) + r), n);
    const unsigned long long v = 1ull << (mix32(i + r) & 31u) | (1ull << 40);   // low bits: OR pattern, bit 40+: count via add
    if (MODE == 1) __hip_atomic_fetch_add(&base[idx], (1ull << 40) , __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_fetch_add(&base[idx], (1ull << 40), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    (void)v;
  }
}
__global__ void reduce(const unsigned long long* tab, uint32_t n, int copies, unsigned long long* total) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  unsigned long long s = 0;
  for (int c = 0; c < copies; ++c) s += tab[(size_t)c * n + i] >> 40;
  atomicAdd(total, s);
}
template <int MODE>
void run(const char* name, unsigned long long* tab, uint32_t n, uint32_t nthreads, uint32_t* xhist, unsigned long long* total) {
  const int copies = MODE == 0 ? 1 : 8;
  hipMemset(tab, 0, (size_t)copies * n * 8); hipMemset(total, 0, 8); hipMemset(xhist, 0, 8 * 16 * 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int reps = 10;
  hipEventRecord(a, 0);
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((k<MODE>), dim3(nthreads / 256), dim3(256), 0, 0, tab, n, 7 + w, xhist);
  hipEventRecord(b, 0); hipEventSynchronize(b);
  float ms = 0; hipEventElapsedTime(&ms, a, b);
  hipLaunchKernelGGL(reduce, dim3((n + 255) / 256), dim3(256), 0, 0, tab, n, copies, total);
  unsigned long long t = 0; hipMemcpy(&t, total, 8, hipMemcpyDeviceToHost);
  const unsigned long long want = (unsigned long long)nthreads * 3 * reps;
  printf("%-34s %8.1f G atomics/s   total %llu / %llu  %s\n", name, (double)want / (ms * 1e-3) / 1e9, t, want, t == want ? "EXACT" : "LOST UPDATES");
}
int main() {
  const uint32_t nthreads = 1u << 20, n = 1u << 20;   // 1 M words of 8 B = 8 MB per copy
  unsigned long long* tab; uint32_t* xhist; unsigned long long* total;
  hipMalloc(&tab, (size_t)8 * n * 8); hipMalloc(&xhist, 8 * 16 * 4); hipMalloc(&total, 8);
  run<0>("agent scope, one table", tab, n, nthreads, xhist, total);
  run<1>("workgroup scope, per-XCD copies", tab, n, nthreads, xhist, total);
  run<2>("agent scope, per-XCD copies", tab, n, nthreads, xhist, total);
  std::vector<uint32_t> h(8 * 16); hipMemcpy(h.data(), xhist, h.size() * 4, hipMemcpyDeviceToHost);
  printf("blocks per (xcc_id, blockIdx%%8):\n");
  for (int x = 0; x < 8; ++x) { printf("  xcc %d:", x); for (int b = 0; b < 8; ++b) printf(" %5u", h[x * 16 + b]); printf("\n"); }
  return 0;
}
