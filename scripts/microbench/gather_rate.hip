// Microbenchmark (measurement tool, not product): request rate of scattered accesses on MI355X as a
// function of table size and access kind.  One thread = one "member" doing R independent accesses at
// hashed indices, like the tick kernels do.  Prints G accesses/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__device__ inline uint32_t mix32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
// kind 0: 64-B line gather as 4 x 16 B; 1: 16-B gather; 2: 4-B gather; 3: 4-B scattered store;
// 4: returning atomicAdd on 4 B; 5: non-returning atomicAdd; 6: atomic + dependent 4-B store (inbox push)
template <int KIND, int R>
__global__ __launch_bounds__(256) void k(uint4* tab, uint32_t* tab2, uint32_t nlines, uint32_t seed, uint32_t* sink) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  uint32_t acc = 0;
  uint32_t idx[R];
#pragma unroll
  for (int r = 0; r < R; ++r) idx[r] = __umulhi(mix32(mix32(i ^ seed) + r), nlines);
  if (KIND == 0) {
#pragma unroll
    for (int r = 0; r < R; ++r) { const uint4* p = tab + (size_t)idx[r] * 4; uint4 a = p[0], b = p[1], c = p[2], d = p[3]; acc += a.x ^ b.y ^ c.z ^ d.w; }
  } else if (KIND == 1) {
#pragma unroll
    for (int r = 0; r < R; ++r) { uint4 a = tab[idx[r]]; acc += a.x ^ a.w; }
  } else if (KIND == 2) {
#pragma unroll
    for (int r = 0; r < R; ++r) acc += tab2[idx[r]];
  } else if (KIND == 3) {
#pragma unroll
    for (int r = 0; r < R; ++r) tab2[idx[r]] = i;
  } else if (KIND == 4) {
    uint32_t o[R];
#pragma unroll
    for (int r = 0; r < R; ++r) o[r] = atomicAdd(&tab2[idx[r]], 1u);
#pragma unroll
    for (int r = 0; r < R; ++r) acc += o[r];
  } else if (KIND == 5) {
#pragma unroll
    for (int r = 0; r < R; ++r) atomicAdd(&tab2[idx[r]], 1u);
  } else if (KIND == 6) {
    uint32_t o[R];
#pragma unroll
    for (int r = 0; r < R; ++r) o[r] = atomicAdd(&tab2[idx[r]], 1u);
    uint32_t* rows = reinterpret_cast<uint32_t*>(tab);
#pragma unroll
    for (int r = 0; r < R; ++r) rows[(size_t)idx[r] * 16 + (o[r] & 15u)] = i;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
template <int KIND, int R>
double run(uint4* tab, uint32_t* tab2, uint32_t nlines, uint32_t nthreads, uint32_t* sink) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<KIND, R>), dim3(nthreads / 256), dim3(256), 0, 0, tab, tab2, nlines, 100 + w, sink);
  hipEventRecord(a, 0);
  const int reps = 20;
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((k<KIND, R>), dim3(nthreads / 256), dim3(256), 0, 0, tab, tab2, nlines, 7 + w, sink);
  hipEventRecord(b, 0); hipEventSynchronize(b);
  float ms = 0; hipEventElapsedTime(&ms, a, b);
  return (double)nthreads * R * reps / (ms * 1e-3) / 1e9;
}
int main() {
  const uint32_t nthreads = 1u << 20;
  uint4* tab; uint32_t* tab2; uint32_t* sink;
  const size_t maxbytes = (size_t)4 << 30;
  hipMalloc(&tab, maxbytes); hipMemset(tab, 1, maxbytes);
  hipMalloc(&tab2, (size_t)1 << 30); hipMemset(tab2, 0, (size_t)1 << 30);
  hipMalloc(&sink, 64);
  const char* names[] = {"gather64B(4x16)", "gather16B", "gather4B", "store4B", "atomic_ret", "atomic_noret", "atomic+store(inbox)"};
  printf("threads=%u R=6 accesses each; G accesses/s by table size\n", nthreads);
  printf("%-22s", "table"); for (int kd = 0; kd < 7; ++kd) printf("%20s", names[kd]); printf("\n");
  for (size_t mb : {1, 4, 16, 64, 256, 1024, 4096}) {
    const size_t bytes = mb << 20;
    const uint32_t nl64 = (uint32_t)(bytes / 64), nl16 = (uint32_t)(bytes / 16), nw = (uint32_t)(std::min(bytes, (size_t)1 << 30) / 4);
    printf("%6zu MB             ", mb);
    printf("%20.1f", run<0, 6>(tab, tab2, nl64, nthreads, sink));
    printf("%20.1f", run<1, 6>(tab, tab2, nl16, nthreads, sink));
    printf("%20.1f", run<2, 6>(tab, tab2, nw, nthreads, sink));
    printf("%20.1f", run<3, 6>(tab, tab2, nw, nthreads, sink));
    printf("%20.1f", run<4, 6>(tab, tab2, nw, nthreads, sink));
    printf("%20.1f", run<5, 6>(tab, tab2, nw, nthreads, sink));
    printf("%20.1f", run<6, 3>(tab, tab2, std::min(nl64, nw), nthreads, sink));
    printf("\n"); fflush(stdout);
  }
  return 0;
}
