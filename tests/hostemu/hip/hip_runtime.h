// TEST INFRASTRUCTURE, NOT PRODUCT.
//
// A minimal stand-in for <hip/hip_runtime.h> that lets g++ compile the product's kernel sources
// (swim_amd/csrc/*.hip, *.h) UNCHANGED into a host library, tests/hostemu/_build/libswimsim_hostemu.so,
// so that the kernels' LOGIC can be parity-checked against the CPU oracle on machines without a GPU
// (this container has none; GPU minutes are rationed).  Only tests/ loads that library; the product
// loader (swim_amd/_lib.py) knows nothing about it and libswimsim.so has no CPU path.
//
// Execution model: blocks run one after another; the threads of a block are ucontext fibres run
// round-robin, switching at __syncthreads() / wave intrinsics.  A fibre waiting in __syncthreads() is
// released only when every live fibre of the block waits there; a fibre inside a wave intrinsic is released
// after one scheduler round (its wave's lanes have all deposited their values by then), so waves may run
// loops of different lengths between two block barriers, as on the GPU.  Lanes that do not take part in a
// wave intrinsic contribute 0.  That reproduces barrier semantics and LDS sharing exactly, and makes
// every run deterministic; it does NOT reproduce the GPU's memory
// model, scheduling or performance -- the `-m gpu` tests remain the parity tests proper.
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <cstdio>
#include <chrono>
#include <tuple>
#include <utility>
#include <vector>

#define __host__
#define __device__
#define __global__
#define __shared__ static
#define __launch_bounds__(...)
#define __forceinline__ inline
#define __restrict__

struct uint2 { uint32_t x, y; };
struct uint3 { uint32_t x, y, z; };
struct alignas(16) uint4 { uint32_t x, y, z, w; };
struct alignas(16) ulonglong2 { unsigned long long x, y; };
struct alignas(32) ulonglong4 { unsigned long long x, y, z, w; };
static inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{x, y}; }
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
static inline ulonglong2 make_ulonglong2(unsigned long long x, unsigned long long y) { return ulonglong2{x, y}; }
static inline ulonglong4 make_ulonglong4(unsigned long long x, unsigned long long y, unsigned long long z, unsigned long long w) { return ulonglong4{x, y, z, w}; }
struct dim3 {
  uint32_t x, y, z;
  dim3(uint32_t x_ = 1, uint32_t y_ = 1, uint32_t z_ = 1) : x(x_), y(y_), z(z_) {}
};

using std::max;
using std::min;

// ---- error / runtime API (single "device", synchronous) ----------------------------------------
typedef enum hipError_t { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorInvalidValue = 1, hipErrorPeerAccessAlreadyEnabled = 704 } hipError_t;
typedef struct hostemu_stream* hipStream_t;
typedef struct hostemu_event { std::chrono::steady_clock::time_point t; }* hipEvent_t;
enum { hipStreamNonBlocking = 1 };
typedef enum { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 } hipMemcpyKind;

static inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "ok" : "hostemu error"; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipMalloc(void** p, size_t bytes) {
#ifdef HOSTEMU_EXACT_ALLOC            // (AddressSanitizer runs: no slack behind an allocation, so that an access one element past it is caught)
  if (posix_memalign(p, 256, bytes ? bytes : 1)) *p = nullptr;
#else
  *p = aligned_alloc(256, (bytes + 255) & ~(size_t)255);
#endif
  return *p ? hipSuccess : hipErrorOutOfMemory;
}
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipHostMalloc(void** p, size_t bytes, unsigned = 0) { *p = malloc(bytes); return *p ? hipSuccess : hipErrorOutOfMemory; }
static inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipMemset(void* p, int v, size_t n) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpy2D(void* d, size_t dpitch, const void* s, size_t spitch, size_t width, size_t height, hipMemcpyKind) {
  for (size_t r = 0; r < height; ++r) memcpy((char*)d + r * dpitch, (const char*)s + r * spitch, width);
  return hipSuccess;
}
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyPeerAsync(void* d, int, const void* s, int, size_t n, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (hipStream_t)(uintptr_t)1; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new hostemu_event(); return hipSuccess; }
enum { hipEventDisableTiming = 2 };
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new hostemu_event(); return hipSuccess; }
static inline hipError_t hipDeviceCanAccessPeer(int* can, int, int) { *can = 1; return hipSuccess; }
static inline hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
// graphs (the SWIMSIM_GRAPH measurement knob): launches run at once here, a graph is nothing
typedef void* hipGraph_t; typedef void* hipGraphExec_t; typedef void* hipGraphNode_t;
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal, hipStreamCaptureModeThreadLocal, hipStreamCaptureModeRelaxed };
static inline hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return hipSuccess; }
static inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) { *g = nullptr; return hipSuccess; }
static inline hipError_t hipGraphInstantiate(hipGraphExec_t* e, hipGraph_t, hipGraphNode_t*, char*, size_t) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }
static inline hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
  *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
  return hipSuccess;
}

// ---- the fibre scheduler -------------------------------------------------------------------------
namespace hostemu {
constexpr size_t STACK_BYTES = 512 * 1024;
// A fibre switch = the callee-saved registers and the stack pointer (x86-64 System V): ~10 ns.  glibc's swapcontext also
// saves and restores the signal mask -- a system call per switch, and a wave intrinsic is two switches per lane: the
// emulated suites spent most of their time there.
#if !defined(__x86_64__)
#error "tests/hostemu: the fibre switch is written for x86-64"
#endif
__attribute__((naked, noinline)) static void fibre_switch(void** /*save_sp: rdi*/, void* /*load_sp: rsi*/) {
  asm volatile(
      "pushq %rbp\n\tpushq %rbx\n\tpushq %r12\n\tpushq %r13\n\tpushq %r14\n\tpushq %r15\n\t"
      "movq %rsp, (%rdi)\n\t"
      "movq %rsi, %rsp\n\t"
      "popq %r15\n\tpopq %r14\n\tpopq %r13\n\tpopq %r12\n\tpopq %rbx\n\tpopq %rbp\n\t"
      "ret\n\t");
}
struct Fibre { void* sp; void* stack; int state; };   // 0 runnable, 1 at block barrier, 2 done, 3 in a wave exchange
struct Sched {
  void* main_sp = nullptr;
  std::vector<Fibre> f;
  uint32_t nthreads = 0, cur = 0;
  void (*entry)(void*) = nullptr; void* arg = nullptr;
  uint64_t exch[2][1024];                                   // cross-lane exchange (shfl / ballot / dpp), two alternating
  uint8_t par[1024];                                        // buffers: one yield per operation is enough (a lane still
                                                            // reading buffer A cannot be overtaken by a write to A: the
                                                            // writer would have to pass the next operation's yield first)
};
inline Sched& sched() { static Sched s; return s; }
inline dim3& tidx() { static dim3 v; return v; }
inline dim3& bidx() { static dim3 v; return v; }
inline dim3& bdim() { static dim3 v; return v; }
inline dim3& gdim() { static dim3 v; return v; }

inline void barrier() {
  Sched& s = sched();
  Fibre& me = s.f[s.cur];
  me.state = 1;
  fibre_switch(&me.sp, s.main_sp);
}
inline void wave_yield() {
  Sched& s = sched();
  Fibre& me = s.f[s.cur];
  me.state = 3;
  fibre_switch(&me.sp, s.main_sp);
}
inline void trampoline() {
  Sched& s = sched();
  s.entry(s.arg);
  s.f[s.cur].state = 2;
  fibre_switch(&s.f[s.cur].sp, s.main_sp);
  abort();                                                 // a finished fibre is never resumed
}
// run one block of nthreads fibres to completion
inline void run_block(uint32_t nthreads, void (*entry)(void*), void* arg) {
  Sched& s = sched();
  if (s.f.size() < nthreads) {
    size_t old = s.f.size();
    s.f.resize(nthreads);
    for (size_t k = old; k < nthreads; ++k) s.f[k].stack = malloc(STACK_BYTES);
  }
  s.nthreads = nthreads; s.entry = entry; s.arg = arg;
  memset(s.par, 0, sizeof s.par);
  for (uint32_t k = 0; k < nthreads; ++k) {
    // a fresh stack as fibre_switch expects to find a suspended one: six register slots, then the address it "returns"
    // to (the trampoline, entered with the stack pointer 8 modulo 16 as after a call), then a null return address
    uintptr_t top = ((uintptr_t)s.f[k].stack + STACK_BYTES) & ~(uintptr_t)15;
    void** sp = reinterpret_cast<void**>(top);
    *--sp = nullptr;
    *--sp = reinterpret_cast<void*>(&trampoline);
    for (int r = 0; r < 6; ++r) *--sp = nullptr;
    s.f[k].sp = sp;
    s.f[k].state = 0;
  }
  for (;;) {
    uint32_t live = 0;
    for (uint32_t k = 0; k < nthreads; ++k) {
      if (s.f[k].state != 0) continue;
      s.cur = k;
      tidx() = dim3(k, 0, 0);
      fibre_switch(&s.main_sp, s.f[k].sp);
    }
    uint32_t in_wave = 0;
    for (uint32_t k = 0; k < nthreads; ++k) if (s.f[k].state == 3) { s.f[k].state = 0; in_wave++; live++; }
    if (!in_wave)                                            // the block barrier opens when nobody is on the way to it
      for (uint32_t k = 0; k < nthreads; ++k) if (s.f[k].state == 1) { s.f[k].state = 0; live++; }
    if (!live) break;                                        // all done (exited threads leave barriers)
  }
}

template <typename F, typename Tuple, size_t... I>
void call_with(F f, Tuple& t, std::index_sequence<I...>) { f(std::get<I>(t)...); }

template <typename... KArgs, typename... Args>
void launch(void (*kernel)(KArgs...), dim3 grid, dim3 block, Args... args) {
  std::tuple<KArgs...> pack(static_cast<KArgs>(args)...);
  struct Ctx { void (*k)(KArgs...); std::tuple<KArgs...>* p; } ctx{kernel, &pack};
  bdim() = block; gdim() = grid;
  for (uint32_t b = 0; b < grid.x; ++b) {
    bidx() = dim3(b, 0, 0);
    run_block(block.x, [](void* a) {
      Ctx* c = static_cast<Ctx*>(a);
      call_with(c->k, *c->p, std::index_sequence_for<KArgs...>{});
    }, &ctx);
  }
}
// ---- path statistics (-DSWIM_PATH_STATS): how often each marked site of a kernel runs, per lane and per wave
// (a wave executes a site max-over-its-lanes times): the divergence profile of DESIGN.md section 5
constexpr int PSTAT_SITES = 48;
struct PStat { std::vector<uint32_t> tab; uint32_t nthreads = 0; bool on = false; };
inline PStat& pstat_state() { static PStat p; return p; }
inline void pstat(int site, uint32_t n = 1) {
  PStat& p = pstat_state();
  if (!p.on) return;
  const uint32_t g = bidx().x * bdim().x + tidx().x;
  if (g >= p.nthreads) return;
  p.tab[(size_t)site * p.nthreads + g] += n;
}
}  // namespace hostemu
extern "C" __attribute__((used, visibility("default"))) void hostemu_pstat_begin(uint32_t nthreads) {
  hostemu::PStat& p = hostemu::pstat_state();
  p.nthreads = nthreads; p.tab.assign((size_t)hostemu::PSTAT_SITES * nthreads, 0u); p.on = true;
}
// out[site] = {sum over lanes, sum over waves of the max over lanes, waves with a non-zero lane}
extern "C" __attribute__((used, visibility("default"))) void hostemu_pstat_end(uint64_t* out) {
  hostemu::PStat& p = hostemu::pstat_state();
  for (int s = 0; s < hostemu::PSTAT_SITES; ++s) {
    uint64_t sum = 0, wsum = 0, wn = 0;
    for (uint32_t w = 0; w < p.nthreads; w += 64) {
      uint32_t mx = 0;
      for (uint32_t l = w; l < w + 64 && l < p.nthreads; ++l) { const uint32_t v = p.tab[(size_t)s * p.nthreads + l]; sum += v; mx = v > mx ? v : mx; }
      wsum += mx; wn += mx ? 1 : 0;
    }
    out[3 * s] = sum; out[3 * s + 1] = wsum; out[3 * s + 2] = wn;
  }
  p.on = false;
}
#ifdef SWIM_PATH_STATS
#define PSTAT(...) hostemu::pstat(__VA_ARGS__)
#endif

#define threadIdx (hostemu::tidx())
#define blockIdx (hostemu::bidx())
#define blockDim (hostemu::bdim())
#define gridDim (hostemu::gdim())
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) hostemu::launch(kernel, grid, block, __VA_ARGS__)

static inline void __syncthreads() { hostemu::barrier(); }
static inline void __threadfence() {}
static inline void __threadfence_block() {}
static inline uint32_t __umulhi(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
static inline int __popc(uint32_t x) { return __builtin_popcount(x); }
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __ffs(uint32_t x) { return __builtin_ffs((int)x); }
static inline int __ffsll(unsigned long long x) { return __builtin_ffsll((long long)x); }
static inline int __clz(uint32_t x) { return x ? __builtin_clz(x) : 32; }

// cross-lane operations: every thread of the block must call them convergently (as on the GPU, where
// the product code only uses them at wave-uniform points)
template <typename T>
static inline T __shfl_down(T x, int delta, int width = 64) {
  auto& s = hostemu::sched();
  const uint32_t tid = hostemu::tidx().x;
  uint64_t v = 0; memcpy(&v, &x, sizeof(T) < 8 ? sizeof(T) : 8);
  uint64_t* xb = s.exch[s.par[tid] ^= 1];
  xb[tid] = v;
  hostemu::wave_yield();
  const uint32_t lane = tid % (uint32_t)width, src = lane + (uint32_t)delta < (uint32_t)width ? tid + (uint32_t)delta : tid;
  uint64_t r = src < s.nthreads ? xb[src] : v;
  T out; memcpy(&out, &r, sizeof(T) < 8 ? sizeof(T) : 8);
  return out;
}
template <typename T>
static inline T __shfl(T x, int srcLane, int width = 64) {
  auto& s = hostemu::sched();
  const uint32_t tid = hostemu::tidx().x;
  uint64_t v = 0; memcpy(&v, &x, sizeof(T) < 8 ? sizeof(T) : 8);
  uint64_t* xb = s.exch[s.par[tid] ^= 1];
  xb[tid] = v;
  hostemu::wave_yield();
  const uint32_t base = tid - tid % (uint32_t)width, src = base + ((uint32_t)srcLane % (uint32_t)width);
  uint64_t r = src < s.nthreads ? xb[src] : v;
  T out; memcpy(&out, &r, sizeof(T) < 8 ? sizeof(T) : 8);
  return out;
}
static inline unsigned long long __ballot(int pred) {
  auto& s = hostemu::sched();
  const uint32_t tid = hostemu::tidx().x;
  uint64_t* xb = s.exch[s.par[tid] ^= 1];
  xb[tid] = pred ? 1u : 0u;
  hostemu::wave_yield();
  const uint32_t base = tid - tid % 64u;
  unsigned long long m = 0;
  for (uint32_t l = 0; l < 64u && base + l < s.nthreads; ++l) if (xb[base + l]) m |= 1ull << l;
  return m;
}

// DPP / lane-read builtins the kernels use for wave reductions (swim_kernels.h wave_sum): gfx9 semantics of
// v_mov_b32_dpp for row_shr:n (0x110 + n), row_bcast:15 (0x142), row_bcast:31 (0x143); a lane whose source is
// invalid or whose row / bank is masked off keeps `old`
static inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
  auto& s = hostemu::sched();
  const uint32_t tid = hostemu::tidx().x;
  uint64_t* xb = s.exch[s.par[tid] ^= 1];
  xb[tid] = (uint32_t)src;
  hostemu::wave_yield();
  const uint32_t base = tid - tid % 64u, lane = tid % 64u, row = lane / 16u, inrow = lane % 16u;
  int out = old;
  if (((row_mask >> row) & 1) && ((bank_mask >> (inrow / 4u)) & 1)) {
    int srcl = -1;
    if (ctrl >= 0x111 && ctrl <= 0x11F) { const uint32_t n = (uint32_t)ctrl - 0x110u; if (inrow >= n) srcl = (int)(lane - n); }
    else if (ctrl == 0x142) { if (row >= 1) srcl = (int)(row * 16u - 1u); }
    else if (ctrl == 0x143) { if (row >= 2) srcl = 31; }
    else abort();
    if (srcl >= 0) out = (int)(uint32_t)xb[base + (uint32_t)srcl];
    else if (bound_ctrl) out = 0;
  }
  return out;
}
static inline int __builtin_amdgcn_readlane(int v, int lane) {
  auto& s = hostemu::sched();
  const uint32_t tid = hostemu::tidx().x;
  uint64_t* xb = s.exch[s.par[tid] ^= 1];
  xb[tid] = (uint32_t)v;
  hostemu::wave_yield();
  return (int)(uint32_t)xb[tid - tid % 64u + (uint32_t)lane];
}
// s_waitcnt: nothing is in flight here; s_barrier: the block barrier; wave_barrier: lanes are fibres, a yield
// lets every lane of the wave reach this point first
static inline void __builtin_amdgcn_s_waitcnt(int) {}
#define __builtin_amdgcn_fence(...) ((void)0)
static inline void __builtin_amdgcn_s_barrier() { hostemu::barrier(); }
static inline void __builtin_amdgcn_wave_barrier() { hostemu::wave_yield(); }

// ---- atomics (fibres never run concurrently: plain read-modify-write is atomic) ----------------------
template <typename T> static inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
static inline unsigned atomicAdd(unsigned* p, int v) { unsigned o = *p; *p = o + (unsigned)v; return o; }
template <typename T> static inline T atomicOr(T* p, T v) { T o = *p; *p = o | v; return o; }
template <typename T> static inline T atomicAnd(T* p, T v) { T o = *p; *p = o & v; return o; }
template <typename T> static inline T atomicXor(T* p, T v) { T o = *p; *p = o ^ v; return o; }
template <typename T> static inline T atomicMin(T* p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <typename T> static inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <typename T> static inline T atomicCAS(T* p, T cmp, T v) { T o = *p; if (o == cmp) *p = v; return o; }
template <typename T> static inline T atomicExch(T* p, T v) { T o = *p; *p = v; return o; }
