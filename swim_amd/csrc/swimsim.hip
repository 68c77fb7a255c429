// swimsim.hip -- C ABI (include/swimsim.h) over the gfx950 tick kernels.
//
// Host side of libswimsim.so: owns device memory, the fault schedule and the launch
// sequence (per tick: begin_kernel -> probe_kernel -> merge_kernel on one HIP stream; no host
// synchronisation inside swimsim_step).
// There is no CPU implementation behind this ABI: without a HIP device swimsim_create fails.
#include "../../include/swimsim.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "swim_kernels.h"
#include "swim_sparse.h"

using namespace swim;

namespace {

constexpr uint32_t INJECT_CAP = 4096;   // rumours from outside the simulation per tick (swimsim_inject_rumor)

struct Fault { uint32_t tick, member, up, order; };

thread_local std::string g_create_err;

uint32_t ceil_log2(uint64_t x) { uint32_t r = 0; while ((1ull << r) < x) r++; return r; }

}  // namespace

struct swimsim {
  swimsim_config_t cfg{};
  DevState d{};
  int device = 0;
  hipStream_t stream = nullptr;
  uint64_t tick = 0;
  std::vector<Fault> faults;      // sorted by (tick, order)
  uint32_t fault_order = 0;
  FaultRec* d_faults = nullptr; size_t d_faults_cap = 0;
  DevState* d_state = nullptr;                 // device copy of d (SWIM_STATE_BY_POINTER builds)
  uint32_t* d_joined = nullptr;                // members that came up in the tick being applied (as many as fault records)
  unsigned long long* d_scratch64 = nullptr;   // digest accumulator
  uint32_t* d_sel = nullptr;                   // [0..255] picks, [256] count, [257..] excludes
  size_t d_sel_cap = 0;
  std::vector<void*> allocs;
  std::vector<swimsim_event_t> host_events;    // drained from the device ring, not yet handed out
  bool timing = false;                         // HIP-event timing of the tick kernels
  std::vector<hipEvent_t> ev_pool;             // 3 events per tick: before probe, between, after merge
  double probe_ms = 0, merge_ms = 0; uint64_t timed_ticks = 0;
  double graph_ms = 0;                      // SWIMSIM_GRAPH=1: launch-to-completion time of the last step's graph
  bool poisoned = false;
  std::string err;
  bool recs_busy = false;                      // the last tick looked at allocated more rumour ids than the masks tolerate: records_kernel_every_tick
  std::vector<InjectRec> injections;           // swimsim_inject_rumor: delivered in the next tick stepped
  InjectRec* d_inject = nullptr;
  std::vector<swimsim_view_entry_t> settled_alive;   // settled subjects that stay listed (Alive at i > 0), as of settled_alive_tick
  uint64_t settled_alive_tick = ~0ull;
  // sharded stepping (swimsim_shard_*): which phase of the current tick comes next, the tick's fault slice
  int shard_phase = 0;
  bool begun = false;                          // swimsim_shard_phase0 applied this tick's faults already (part A of begin_kernel)
  size_t begun_fend = 0;
  bool tick_inj = false;                       // messages from outside went into this tick's inboxes (begin_kernel's part bit 2)
  uint32_t j_in[MAX_SHARDS] = {};              // join-pull records received in round 0
  uint32_t* h_sync = nullptr;                  // pinned host copy of the globals (flags + send counts)
  hipEvent_t tick_ev[3] = {nullptr, nullptr, nullptr};
  uint32_t sp_grid_probe = 0, sp_grid_merge = 0;   // bounded member maps: workgroups of the two tick kernels
  bool fold_begin = true;                          // plain ticks run without begin_kernel (SWIMSIM_FOLD_BEGIN=0 at create: never)
  bool sp_probe_by_wave = false;                   // SWIMSIM_SP_PROBE=wave at create: the wave-per-member probe kernel (A/B, tests)
};

// the tick kernels' state argument: by value, or (-DSWIM_STATE_BY_POINTER, measurement knob) a pointer to a device copy
#ifdef SWIM_STATE_BY_POINTER
#define SWIM_STATE_ARG(h) ((const swim::DevState*)(h)->d_state)
#else
#define SWIM_STATE_ARG(h) (h)->d
#endif

namespace {

int set_err(swimsim* h, int code, const std::string& msg) {
  if (h) { h->err = msg; if (code == SWIMSIM_ERR_CAPACITY) h->poisoned = true; }
  else g_create_err = msg;
  return code;
}

#define HIPCHK(h, expr)                                                                         \
  do {                                                                                          \
    hipError_t e_ = (expr);                                                                     \
    if (e_ != hipSuccess)                                                                       \
      return set_err((h), e_ == hipErrorOutOfMemory ? SWIMSIM_ERR_NOMEM : SWIMSIM_ERR_DEVICE,   \
                     std::string(#expr) + ": " + hipGetErrorString(e_));                        \
  } while (0)

template <typename T>
int dev_alloc(swimsim* h, T** p, size_t count, int fill_byte) {
  void* q = nullptr;
  size_t bytes = count * sizeof(T);
  if (bytes == 0) bytes = sizeof(T);
  HIPCHK(h, hipMalloc(&q, bytes));
  h->allocs.push_back(q);
  HIPCHK(h, hipMemsetAsync(q, fill_byte, bytes, h->stream));
  *p = static_cast<T*>(q);
  return SWIMSIM_OK;
}

int resolve_config(const swimsim_config_t* in, swimsim_config_t* c, std::string* err) {
  if (!in) { *err = "config is NULL"; return SWIMSIM_ERR_INVALID; }
  if (in->struct_size != sizeof *in || in->abi_version != SWIMSIM_ABI_VERSION) {
    *err = "config struct_size/abi_version mismatch"; return SWIMSIM_ERR_INVALID; }
  *c = *in;
  if (c->n_members < 2 || c->n_members > 0x7FFFFFFFu) { *err = "n_members must be in [2, 2^31)"; return SWIMSIM_ERR_INVALID; }
  if (c->num_to_gossip < 0) { *err = "num_to_gossip must be >= 0"; return SWIMSIM_ERR_INVALID; }
  if (c->probes_per_tick == 0) c->probes_per_tick = c->num_to_gossip;
  if (c->indirect_k == 0) c->indirect_k = c->num_to_gossip;
  if (c->probes_per_tick < 0 || c->probes_per_tick > 16 || c->indirect_k < 0 || c->indirect_k > 16) {
    *err = "probes_per_tick / indirect_k must be in [0,16]"; return SWIMSIM_ERR_INVALID; }
  if (c->loss_ppm > 1000000u) { *err = "loss_ppm must be <= 1000000"; return SWIMSIM_ERR_INVALID; }
  if (c->suspicion_ticks == 0) c->suspicion_ticks = 3 * ceil_log2(c->n_members);
  if (c->suspicion_ticks == 0) c->suspicion_ticks = 1;
  if (c->retransmit_mult == 0) c->retransmit_mult = 3;
  if ((uint64_t)c->retransmit_mult * ceil_log2((uint64_t)c->n_members + 1) > 255) {
    *err = "retransmit budget exceeds 255"; return SWIMSIM_ERR_INVALID; }
  if (c->max_subjects == 0) {
    // enough view columns for the subjects of ~256 periods of false suspicions (all in parts per million, integers only: the oracle and the product must agree on the number), within 32 GB of columns
    uint64_t q = 1000000u - c->loss_ppm, q2 = q * q / 1000000u, q4 = q2 * q2 / 1000000u;
    uint64_t pf = 1000000u - q2;                                   // the direct probe fails ...
    for (int k = 0; k < c->indirect_k; k++) pf = pf * (1000000u - q4) / 1000000u;   // ... and every proxy chain
    uint64_t est = pf * (uint64_t)c->probes_per_tick * c->n_members / 1000000u * 512u + 1024u;
    uint64_t mem = 32000000000ull / (8ull * c->n_members);
    if (est > mem) est = mem < 64 ? 64 : mem;
    if (est > 60000u) est = 60000u;
    if (est > c->n_members) est = c->n_members;
    c->max_subjects = (uint32_t)est;
  }
  if (c->max_subjects > 60000u) { *err = "max_subjects must be <= 60000"; return SWIMSIM_ERR_INVALID; }
  {
    const uint32_t gmin = c->suspicion_ticks + c->retransmit_mult * ceil_log2((uint64_t)c->n_members + 1) + 2;
    if (c->gc_ticks == SWIMSIM_GC_AUTO) c->gc_ticks = gmin;
    if (c->gc_ticks && c->gc_ticks < gmin) {
      *err = "gc_ticks must be 0, SWIMSIM_GC_AUTO or >= suspicion_ticks + L + 2 = " + std::to_string(gmin); return SWIMSIM_ERR_INVALID; }
  }
  if (c->event_cap == 0) c->event_cap = 1u << 20;
  if (c->event_mask == 0) c->event_mask = SWIMSIM_EVMASK_DEFAULT;
  if (c->inbox_cap == 0) {
    // expected deliveries per member-tick if every message carried a payload:
    // P pings in + P acks in + 4 hops per proxied probe
    const double l = c->loss_ppm / 1e6, pf = 1.0 - (1.0 - l) * (1.0 - l);
    const double lam = 2.0 * c->probes_per_tick + 4.0 * c->probes_per_tick * c->indirect_k * pf;
    uint32_t cap = lam <= 8.0 ? 16u : (uint32_t)(lam + 6.0 * std::sqrt(lam) + 8.0);
    c->inbox_cap = std::min<uint32_t>(1024u, (cap + 15u) & ~15u);
  }
  if (c->inbox_cap > 4096u) { *err = "inbox_cap must be <= 4096"; return SWIMSIM_ERR_INVALID; }
  if (c->n_shards == 0) c->n_shards = 1;
  if (c->n_shards > 16 || c->shard_index >= c->n_shards) { *err = "n_shards must be <= 16 and shard_index < n_shards"; return SWIMSIM_ERR_INVALID; }
  if (c->n_members % c->n_shards) { *err = "n_members must be a multiple of n_shards"; return SWIMSIM_ERR_INVALID; }
  if (c->target_scheme > SWIMSIM_TARGETS_ROBUST) { *err = "unknown target_scheme"; return SWIMSIM_ERR_INVALID; }
  if (c->join_pull > 1) { *err = "join_pull must be 0 or 1"; return SWIMSIM_ERR_INVALID; }
  if (c->pull_ticks == 1) { *err = "pull_ticks must be 0 (off) or >= 2"; return SWIMSIM_ERR_INVALID; }
  // (ranges first, combinations after: a value out of range gets the message that says so)
  if (c->push_pull > 1 || (c->push_pull && !c->pull_ticks)) { *err = "push_pull must be 0 or 1 and needs pull_ticks"; return SWIMSIM_ERR_INVALID; }
  if (c->strict_reference_rules > 1) { *err = "strict_reference_rules must be 0 or 1"; return SWIMSIM_ERR_INVALID; }
  if (c->n_shards > 1 && c->n_members > (1u << 27)) { *err = "sharded clusters: n_members must be <= 2^27"; return SWIMSIM_ERR_INVALID; }
  if (c->strict_reference_rules && c->view_cap) {
    *err = "strict_reference_rules cannot be combined with view_cap"; return SWIMSIM_ERR_INVALID; }
  if (c->view_cap) {
    if (c->view_cap < SWIMSIM_VIEW_CAP_MIN || c->view_cap > SWIMSIM_VIEW_CAP_MAX) {
      *err = "view_cap must be 0 (unbounded) or in [" + std::to_string(SWIMSIM_VIEW_CAP_MIN) + ", " + std::to_string(SWIMSIM_VIEW_CAP_MAX) + "]"; return SWIMSIM_ERR_INVALID; }
    if (c->gc_ticks || c->join_pull || c->pull_ticks || c->target_scheme != SWIMSIM_TARGETS_RANDOM) {
      *err = "view_cap (bounded member maps) cannot be combined with gc_ticks, join_pull, pull_ticks or the robust target scheme"; return SWIMSIM_ERR_INVALID; }
  }
  return SWIMSIM_OK;
}

int check_device_errors(swimsim* h) {
  uint32_t g[G_WORDS];
  HIPCHK(h, hipMemcpy(g, h->d.g, sizeof g, hipMemcpyDeviceToHost));
  // (with hysteresis: on above the masks' slack, off again below half of it -- the id rate of a regime fluctuates by a few per tick)
  { const uint32_t ids = g[G_HEAD] - g[G_PREV];
    if (h->d.C || h->d.n_shards != 1) h->recs_busy = false;
    else if (ids > MASK_SLACK) h->recs_busy = true;
    else if (ids <= MASK_SLACK / 2u) h->recs_busy = false; }
  if (g[G_ERR]) {
    std::string m = "capacity exceeded:";
    if (g[G_ERR] & ERRF_SUBJECTS) m += h->d.C ? " subjects-a-member-hears-of-in-one-tick (bounded member maps: the per-tick working set)" : " max_subjects";
    if (g[G_ERR] & ERRF_ROWS) m += " view-rows-in-transit (settled rows wait two ticks before reuse)";
    if (g[G_ERR] & ERRF_OVF) m += " inbox-overflow-list (" + std::to_string(std::max(g[G_OVF0], g[G_OVF1])) + " entries, room for " + std::to_string(h->d.ovf_cap) + ")";
    if (g[G_ERR] & ERRF_INC) m += " incarnation-bits";
    if (g[G_ERR] & ERRF_XCHG) m += " shard-exchange-buffers";
    if (g[G_ERR] & ERRF_TODO) m += " explicit-record-entries (room for " + std::to_string(h->d.todo_cap) + " per region of the tick's list + " + std::to_string(h->d.todo_spill) + " shared)";
    return set_err(h, SWIMSIM_ERR_CAPACITY, m);
  }
  return SWIMSIM_OK;
}

int pull_events(swimsim* h) {
  uint32_t g[G_WORDS];
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipMemcpy(g, h->d.g, sizeof g, hipMemcpyDeviceToHost));
  uint32_t n = std::min(g[G_EVCUR], h->d.event_cap);
  if (n) {
    std::vector<uint4> raw(n);
    HIPCHK(h, hipMemcpy(raw.data(), h->d.events, (size_t)n * sizeof(uint4), hipMemcpyDeviceToHost));
    h->host_events.reserve(h->host_events.size() + n);
    for (const uint4& r : raw) {
      swimsim_event_t e{};
      e.tick = r.x; e.observer = r.y; e.subject = r.z;
      e.cause = (uint8_t)(r.w & 0xFFu);
      const uint32_t key = r.w >> 8;
      e.state = (uint8_t)(key & 3u); e.incarnation = key >> 2;
      h->host_events.push_back(e);
    }
  }
  if (g[G_EVCUR]) HIPCHK(h, hipMemset(h->d.g + G_EVCUR, 0, sizeof(uint32_t)));
  auto key_of = [](const swimsim_event_t& e) { return (e.incarnation << 2) | e.state; };
  std::sort(h->host_events.begin(), h->host_events.end(), [&](const swimsim_event_t& a, const swimsim_event_t& b) {
    if (a.tick != b.tick) return a.tick < b.tick;
    if (a.observer != b.observer) return a.observer < b.observer;
    if (a.subject != b.subject) return a.subject < b.subject;
    return key_of(a) < key_of(b);
  });
  // collapse (tick, observer, subject) to the final (largest) key
  size_t w = 0;
  for (size_t x = 0; x < h->host_events.size(); ++x) {
    if (x + 1 < h->host_events.size() && h->host_events[x + 1].tick == h->host_events[x].tick &&
        h->host_events[x + 1].observer == h->host_events[x].observer &&
        h->host_events[x + 1].subject == h->host_events[x].subject) continue;
    h->host_events[w++] = h->host_events[x];
  }
  h->host_events.resize(w);
  return SWIMSIM_OK;
}

// The robust scheme's rotations for period t (include/swimsim.h): rounds of
// R = ceil((N-1)/P) periods; round r uses a pseudo-random permutation pi_r of 0..N-2 (a keyed bijection on
// ceil(log2(N-1))-bit words -- xor, odd multiplications, xor-shifts, one addition -- restricted to
// [0, N-1) by cycle walking); probe p of period u of the round has offset 1 + pi_r(u P + p).
uint32_t perm_bits(uint32_t x, uint32_t k1, uint32_t k2, uint32_t bits) {
  const uint32_t mask = bits >= 32 ? 0xFFFFFFFFu : (1u << bits) - 1u, sh = bits / 2 ? bits / 2 : 1;
  x = (x ^ k1) & mask;
  x = (x * 0x9E3779B1u) & mask; x ^= x >> sh;
  x = (x * 0x85EBCA6Bu) & mask; x ^= x >> sh;
  x = (x + k2) & mask;
  x = (x * 0xC2B2AE35u) & mask; x ^= x >> sh;
  return x;
}
Offsets robust_offsets(const swimsim* h, uint32_t t) {
  Offsets off{};
  if (h->d.scheme != SWIMSIM_TARGETS_ROBUST) return off;
  const uint32_t M = h->d.NT - 1, P = std::max(1u, h->d.P), R = (M + P - 1) / P, r = t / R, u = t % R;
  const uint32_t mk = mix32(tick_key(h->cfg.seed, r) ^ 0x524F4255u);
  const uint32_t k1 = hash_mk(mk, 1, 0), k2 = hash_mk(mk, 2, 0), bits = ceil_log2(M);
  for (uint32_t p = 0; p < h->d.P; ++p) {
    const uint64_t k = (uint64_t)u * P + p;
    if (k >= M) continue;
    uint32_t x = (uint32_t)k;
    if (bits) do x = perm_bits(x, k1, k2, bits); while (x >= M);
    off.o[p] = 1u + x;
  }
  return off;
}

// Explicit records are read by records_kernel (its own launch: a kernel boundary costs ~10 us on this chip, profiles/r03i_*)
// on handles whose every tick has them for every member -- message loss (after a burst of rumour ids no queue travels
// as a mask); a lossless handle meets them in a few ticks per hundred and
// reads them in a phase at the start of merge_kernel instead.  Same result either way (tests run both on both).
bool records_kernel_every_tick(const swimsim* h) {
  const char* force = std::getenv("SWIMSIM_RECORDS_KERNEL");             // test / measurement knob: 0 = never, 1 = always
  if (force && (force[0] == '0' || force[0] == '1')) return force[0] == '1';
  // ... or the cluster states more new rumours per tick than the masks tolerate (BASELINE.md row 3(s) as written: 9.5 crashes per tick =
  // ~20 ids per tick against MASK_SLACK = 16): every tick travels as explicit records for every member, as under loss.  Looked at
  // once per swimsim_step call (check_device_errors reads the words anyway); merge 1 139 -> 1 064 us there (profiles/r06g_*)
  return h->cfg.loss_ppm != 0 || h->d.strict || h->recs_busy;   // strict reference rules: every delivery is an explicit record
}

// messages from outside the simulation (swimsim_inject_rumor) go into this tick's inboxes BEFORE the start of the tick: the rows they
// open count as stated in this tick (a row nobody holds anything in settles at the end of it, as in the oracle), the ids they take
// are older than the tick's window head.  *any = explicit records exist already (begin_kernel's part bit 2)
int flush_injections(swimsim* h, uint32_t t, bool* any) {
  *any = false;
  if (h->injections.empty()) return SWIMSIM_OK;
  const uint32_t ni = (uint32_t)h->injections.size();
  HIPCHK(h, hipMemcpyAsync(h->d_inject, h->injections.data(), ni * sizeof(InjectRec), hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(inject_kernel, dim3((ni + 63) / 64), dim3(64), 0, h->stream, h->d, t, h->d_inject, ni);
  HIPCHK(h, hipStreamSynchronize(h->stream));   // the host list is reused
  h->injections.clear();
  *any = true;
  return SWIMSIM_OK;
}

// one period of failureDetector for the handle's members; registers follow the probe / proxy arrays: four sizes (12: the
// reference's default numToGossip = 10, src/Util.hs:48); a shard of a cluster runs the instantiation that knows about remote members
void launch_probe(swimsim* h, uint32_t t, uint32_t tk, uint32_t fold, const CrashList& cl = CrashList{}) {
  const uint32_t pk = std::max(h->d.P, h->d.K);
  const Offsets off = robust_offsets(h, t);
  const dim3 g(h->d.nblocks), b(BLOCK);
#define SWIM_LAUNCH_PROBE(PM) do { if (h->d.n_shards > 1) hipLaunchKernelGGL((probe_kernel<PM, true>), g, b, 0, h->stream, SWIM_STATE_ARG(h), t, tk, off, fold, cl); \
                                   else hipLaunchKernelGGL((probe_kernel<PM, false>), g, b, 0, h->stream, SWIM_STATE_ARG(h), t, tk, off, fold, cl); } while (0)
  if (pk <= 4) SWIM_LAUNCH_PROBE(4);
  else if (pk <= 8) SWIM_LAUNCH_PROBE(8);
  else if (pk <= 12) SWIM_LAUNCH_PROBE(12);
  else SWIM_LAUNCH_PROBE(16);
#undef SWIM_LAUNCH_PROBE
}

// sharded dense handles (swim_kernels.h "cross-shard exchange kernels"): grids of the small kernels around the two rounds
uint32_t publish_grid(const swimsim* h) { return std::max(1u, std::min<uint32_t>(h->d.nblocks, 2048u)); }
uint32_t ingest_grid(const swimsim* h) { return std::max(1u, std::min<uint32_t>(h->d.nblocks, 2048u)); }

void launch_tick(swimsim* h, uint32_t t, uint32_t tk, hipEvent_t* ev, uint32_t fold, const CrashList& cl) {
  if (ev) (void)hipEventRecord(ev[0], h->stream);
  launch_probe(h, t, tk, fold, cl);
  if (ev) (void)hipEventRecord(ev[1], h->stream);
  const bool rk = records_kernel_every_tick(h);
  if (rk) hipLaunchKernelGGL(records_kernel, dim3(h->d.nblocks), dim3(BLOCK), 0, h->stream, h->d, t);
  hipLaunchKernelGGL(merge_kernel, dim3(h->d.nblocks), dim3(BLOCK), 0, h->stream, SWIM_STATE_ARG(h), t, rk ? 0u : 1u, fold);
  if (ev) (void)hipEventRecord(ev[2], h->stream);
}

// bounded member maps: probe (map entries per lane by view_cap) and merge (LDS table by view_cap); every wave steps the members
// w, w + W, ... of its grid of W waves.  Measured at 2 M members (profiles/r04f_*, r04g_*): a grid of exactly the workgroups the
// chip holds at once is 15-25 % SLOWER than 16 384 workgroups (the members' work differs: many short-lived workgroups balance,
// a few long ones wait for the slowest); beyond 65 536 the per-workgroup counter flush shows
void size_sparse_grids(swimsim* h) {
  const uint32_t N = h->d.N;
  h->sp_grid_probe = std::max(1u, std::min<uint32_t>((N + SP_WAVES - 1) / SP_WAVES, 16384u));
  const uint32_t mw = h->d.C <= 128 ? 4u : 2u;      // members per workgroup of the merge kernel (its LDS tables)
  h->sp_grid_merge = std::max(1u, std::min<uint32_t>((N + mw - 1) / mw, 16384u));
  { const char* e = std::getenv("SWIMSIM_SP_PROBE"); h->sp_probe_by_wave = e && e[0] == 'w'; }
  // measurement knob: SWIMSIM_SP_GRID="<probe workgroups>,<merge workgroups>" (0 = keep)
  if (const char* e = std::getenv("SWIMSIM_SP_GRID")) {
    unsigned a = 0, b = 0;
    if (std::sscanf(e, "%u,%u", &a, &b) == 2) {
      if (a) h->sp_grid_probe = std::min<uint32_t>(a, (N + SP_WAVES - 1) / SP_WAVES);
      if (b) h->sp_grid_merge = std::min<uint32_t>(b, (N + 1) / 2);
    }
  }
}
void launch_sparse_probe(swimsim* h, uint32_t t, uint32_t tk) {
  // one member per lane (swim_sparse.h sp_probe_lane_kernel); SWIMSIM_SP_PROBE=wave (read at create): the wave-per-member kernel, for A/B
  if (!h->sp_probe_by_wave) {
    const dim3 g(std::min<uint32_t>((h->d.N + BLOCK - 1) / BLOCK, h->d.nblocks));   // (a counter row per workgroup: the rest by grid stride)
    const uint32_t pk = std::max(h->d.P, h->d.K);
    if (pk <= 4) hipLaunchKernelGGL((sp_probe_lane_kernel<4>), g, dim3(BLOCK), 0, h->stream, h->d, t, tk);
    else if (pk <= 8) hipLaunchKernelGGL((sp_probe_lane_kernel<8>), g, dim3(BLOCK), 0, h->stream, h->d, t, tk);
    else hipLaunchKernelGGL((sp_probe_lane_kernel<16>), g, dim3(BLOCK), 0, h->stream, h->d, t, tk);
    return;
  }
  const dim3 gp(h->sp_grid_probe);
  if (h->d.C <= 64) hipLaunchKernelGGL((sp_probe_kernel<1>), gp, dim3(BLOCK), 0, h->stream, h->d, t, tk);
  else if (h->d.C <= 128) hipLaunchKernelGGL((sp_probe_kernel<2>), gp, dim3(BLOCK), 0, h->stream, h->d, t, tk);
  else hipLaunchKernelGGL((sp_probe_kernel<4>), gp, dim3(BLOCK), 0, h->stream, h->d, t, tk);
}
void launch_sparse_merge(swimsim* h, uint32_t t, uint32_t tk) {
  const dim3 gm(h->sp_grid_merge);
  // the per-tick working set of a member: its map + the subjects it hears of for the first time = 4 x the capacity, a table in LDS
  // (256 / 512 / 1 024 slots; beyond it the rank floor of swim_sparse.h)
#ifdef SWIM_SP_PHYS   // test builds: a tiny working set, so that ordinary ticks overflow it and take the rank-floor retries (view_cap <= SWIM_SP_PHYS / 4)
  if (h->d.C * 4u <= SWIM_SP_PHYS) hipLaunchKernelGGL((sp_merge_kernel<SWIM_SP_PHYS, 1, 4>), gm, dim3(256), 0, h->stream, h->d, t, tk);
  else
#endif
  if (h->d.C <= 64) hipLaunchKernelGGL((sp_merge_kernel<256, 1, 4>), gm, dim3(256), 0, h->stream, h->d, t, tk);
  else if (h->d.C <= 128) hipLaunchKernelGGL((sp_merge_kernel<512, 2, 4>), gm, dim3(256), 0, h->stream, h->d, t, tk);
  else hipLaunchKernelGGL((sp_merge_kernel<1024, 4, 2>), gm, dim3(128), 0, h->stream, h->d, t, tk);
}
void launch_sparse_tick(swimsim* h, uint32_t t, uint32_t tk, hipEvent_t* ev) {
  if (ev) (void)hipEventRecord(ev[0], h->stream);
  launch_sparse_probe(h, t, tk);
  if (ev) (void)hipEventRecord(ev[1], h->stream);
  launch_sparse_merge(h, t, tk);
  if (ev) (void)hipEventRecord(ev[2], h->stream);
}

}  // namespace

extern "C" {

int swimsim_default_config(swimsim_config_t* cfg) {
  if (!cfg) return SWIMSIM_ERR_INVALID;
  std::memset(cfg, 0, sizeof *cfg);
  cfg->struct_size = (uint32_t)sizeof *cfg;
  cfg->abi_version = SWIMSIM_ABI_VERSION;
  cfg->num_to_gossip = 10;           // src/Util.hs:48
  cfg->gossip_interval_us = 200000;  // src/Util.hs:49
  return SWIMSIM_OK;
}

const char* swimsim_last_error(const swimsim_t* h) { return h ? h->err.c_str() : g_create_err.c_str(); }

void swimsim_destroy(swimsim_t* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  for (void* p : h->allocs) (void)hipFree(p);
  if (h->d_faults) (void)hipFree(h->d_faults);
  if (h->d_joined) (void)hipFree(h->d_joined);
  if (h->d_sel) (void)hipFree(h->d_sel);
  for (hipEvent_t e : h->ev_pool) (void)hipEventDestroy(e);
  for (hipEvent_t e : h->tick_ev) if (e) (void)hipEventDestroy(e);
  if (h->h_sync) (void)hipHostFree(h->h_sync);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

int swimsim_create(const swimsim_config_t* cfg, swimsim_t** out) {
  if (!out) return SWIMSIM_ERR_INVALID;
  *out = nullptr;
  swimsim_config_t c;
  std::string e;
  int rc = resolve_config(cfg, &c, &e);
  if (rc) return set_err(nullptr, rc, e);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return set_err(nullptr, SWIMSIM_ERR_DEVICE, "no HIP device visible (libswimsim has no CPU fallback)");
  if (c.device < 0 || c.device >= ndev) return set_err(nullptr, SWIMSIM_ERR_INVALID, "device ordinal out of range");
  swimsim* h = new (std::nothrow) swimsim();
  if (!h) return set_err(nullptr, SWIMSIM_ERR_NOMEM, "out of host memory");
  h->cfg = c; h->device = c.device;
  { const char* e = std::getenv("SWIMSIM_FOLD_BEGIN"); h->fold_begin = !(e && e[0] == '0'); }
  auto bail = [&](int code) { g_create_err = h->err; swimsim_destroy(h); return code; };
#define CK(expr) do { int rc_ = (expr); if (rc_) return bail(rc_); } while (0)
#define HK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { h->err = std::string(#expr) + ": " + hipGetErrorString(e_); return bail(e_ == hipErrorOutOfMemory ? SWIMSIM_ERR_NOMEM : SWIMSIM_ERR_DEVICE); } } while (0)
  HK(hipSetDevice(h->device));
  HK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  DevState& d = h->d;
  const uint32_t NT = c.n_members, N = NT / c.n_shards;   // N = members owned by this handle
  d.N = N; d.NT = NT; d.lo = c.shard_index * N; d.n_shards = c.n_shards; d.shard = c.shard_index;
  d.scheme = c.target_scheme;
  d.join_pull = c.join_pull;
  d.pull_T = c.pull_ticks;
  d.strict = c.strict_reference_rules;
  d.push_pull = c.push_pull;
  d.P = (uint32_t)c.probes_per_tick; d.K = (uint32_t)c.indirect_k; d.S = c.suspicion_ticks;
  d.L = c.retransmit_mult * ceil_log2((uint64_t)NT + 1);
  {
    uint64_t thr = ((uint64_t)c.loss_ppm << 32) / 1000000ull;
    d.loss_thr = thr > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)thr;
  }
  d.R_max = c.max_subjects; d.event_cap = c.event_cap; d.event_mask = c.event_mask;
  d.G = c.gc_ticks;
  // settled rows are cleared by the next merge and reusable the tick after: room for the rows in transit
  // (rows in transit can be as many as were live: a correlated burst settling in one tick while new subjects arrive.  The
  // oracle keeps 2 x max_subjects columns; so does the product while the view stays under 64 GB, else what fits, at least 1024)
  {
    uint32_t transit = 0;
    if (d.G) {
      const uint64_t row_bytes = 8ull * (VTILE ? (uint64_t)(N + VTILE - 1) / VTILE * VTILE : N);
      const uint64_t fit = 64000000000ull / row_bytes;
      transit = (uint32_t)std::min<uint64_t>(d.R_max, std::max<uint64_t>(1024u, fit > d.R_max ? fit - d.R_max : 0u));
    }
    d.R_phys = std::min<uint32_t>(65534u, d.R_max + transit);
  }
  d.nblocks = (N + BLOCK - 1) / BLOCK;
  d.inbox_cap = c.inbox_cap;
  d.ovf_cap = std::max<uint32_t>(1u << 16, N / 8);
  {
    // an inbox smaller than the expected fan-in (tests do that on purpose) sends the excess through the
    // overflow list every tick: room for all of it
    const double l = c.loss_ppm / 1e6, pf = 1.0 - (1.0 - l) * (1.0 - l);
    const double lam = 2.0 * c.probes_per_tick + 4.0 * c.probes_per_tick * c.indirect_k * pf;
    if (lam > c.inbox_cap) d.ovf_cap = (uint32_t)std::min<double>(3.0e8, std::max<double>(d.ovf_cap, 2.0 * lam * N + 65536.0));
    // Under heavy loss the views degrade (false suspicions beat refutations: protocol overload),
    // every prober's choice of targets shrinks to the few members it still holds Alive, and those few receive
    // thousands of deliveries per tick while the average stays ~100 (sharded soak, P = K = 10 at 20 % loss: 40 % of
    // all deliveries of a tick beyond the 256-slot inboxes): room for every expected delivery
    d.ovf_cap = (uint32_t)std::min<double>(3.0e8, std::max<double>(d.ovf_cap, lam * N + 65536.0));
  }
  {
    // the records phase of merge_kernel reserves 8 todo entries per explicit-record source: room for the expected number of sources per
    // member (every delivery, when the masks are off) with headroom; overruns are loud (SWIMSIM_ERR_CAPACITY)
    const double l = c.loss_ppm / 1e6, pf = 1.0 - (1.0 - l) * (1.0 - l);
    const double lam = 2.0 * c.probes_per_tick + 4.0 * c.probes_per_tick * c.indirect_k * pf;
    // (per region: a workgroup reserves from region b mod TODO_REGIONS; small clusters have fewer workgroups than regions)
    const uint32_t nb = (N + BLOCK - 1) / BLOCK, share = std::min<uint32_t>(TODO_REGIONS, nb);
    // (a shard of a small cluster: when the views degrade under heavy loss, everybody's few remaining targets may sit on
    // ONE shard -- the soak: 8 shards of 37 members, P = K = 10, 20 % loss -- so up to 65 536 members count in full)
    const double senders = std::max<double>(N, std::min<double>(NT, 65536.0));
    d.todo_cap = (uint32_t)std::min<double>(3.0e7, (senders * (1.25 * lam + 2.0) * PB_SLOTS) / share + 65536.0);
    // ... and a spill area the regions share, for the workgroups whose members take many times their share: a degraded
    // cluster under heavy loss sends everything to the few members it still holds Alive (the GPU sweep: 4 096 members,
    // P = K = 10, 30 % loss, one workgroup's 256 members receive 4x the average).  Up to 65 536 senders it holds the hard
    // bound -- every probe of every member failing, every hop of every indirect probe delivered --, beyond that a quarter of
    // an average tick
    const double hard = 2.0 * c.probes_per_tick + 4.0 * c.probes_per_tick * c.indirect_k;
    d.todo_spill_at = d.todo_cap * share;
    d.todo_spill = (uint32_t)std::min<double>(4.0e8, senders <= 65536.0 ? senders * hard * PB_SLOTS : senders * (1.25 * lam + 2.0) * PB_SLOTS / 4.0);
  }
  d.C = c.view_cap;
  if (d.C) {
    // bounded member maps (swim_sparse.h): the maps, the queue lines, the inboxes -- none of the view / mask / deadline tables
    d.ord_cap = 0; d.r_cap = d.p_cap = d.x_cap = 0;
    d.R_phys = 0;
    d.sp_ack_cap = std::max(1u, d.P) * (1u + d.K);
    // one wave per member, persistent: enough workgroups to fill the chip several times over (256 CUs)
    size_sparse_grids(h);
    d.nblocks = std::max(h->sp_grid_probe, h->sp_grid_merge);     // one counter row per workgroup of the larger grid
    CK(dev_alloc(h, &d.mb, ((size_t)NT + 3) & ~(size_t)3, 0));
    CK(dev_alloc(h, &d.hot, N, 0));
    CK(dev_alloc(h, &d.sp_tab, (size_t)N * 3 * d.C, 0));
    CK(dev_alloc(h, &d.sp_tab_n, N, 0));
    d.sp_bloom_log2 = d.C <= 64 ? 10u : d.C <= 128 ? 11u : 12u;
    CK(dev_alloc(h, &d.sp_bloom, (size_t)N << (d.sp_bloom_log2 - 5u), 0));
    CK(dev_alloc(h, &d.sp_q, (size_t)2 * N * PB_SLOTS, 0));
    if (d.n_shards > 1) {
      // shards of a bounded cluster (swim_sparse.h): the replica of everybody's queue line, the lists of deliveries to members
      // of other shards, the per-peer segments of 16-byte records {dst, src, -, -} they are routed into (kind 1)
      const double l_ = c.loss_ppm / 1e6, pf_ = 1.0 - (1.0 - l_) * (1.0 - l_);
      const double lam_ = 2.0 * c.probes_per_tick + 4.0 * c.probes_per_tick * c.indirect_k * pf_;   // deliveries per member-tick
      const double remote = (double)N * lam_ * (d.n_shards - 1) / d.n_shards;
      d.sp_ord_cap = (uint32_t)std::min<double>(4.0e8, remote * 1.5 / 64.0 + 8.0 * std::sqrt(remote) + 4096.0);
      d.p_cap = (uint32_t)std::min<double>(4.0e8, remote / (d.n_shards - 1) * 1.5 + 8.0 * std::sqrt(remote) + 4096.0);
      CK(dev_alloc(h, &d.sp_qall, (size_t)NT * PB_SLOTS, 0));
      CK(dev_alloc(h, &d.sp_ord, (size_t)64 * d.sp_ord_cap, 0));
      CK(dev_alloc(h, &d.sp_ord_n, (size_t)64 * 16, 0));
      CK(dev_alloc(h, &d.sp_pin, (size_t)MAX_SHARDS, 0));
      CK(dev_alloc(h, &d.p_send, (size_t)d.n_shards * d.p_cap, 0));
      CK(dev_alloc(h, &d.p_recv, (size_t)d.n_shards * d.p_cap, 0));
    }
    CK(dev_alloc(h, &d.sp_out, N, 0));
    CK(dev_alloc(h, &d.ackfrom, (size_t)N * d.sp_ack_cap, 0));
    CK(dev_alloc(h, &d.fail, (size_t)N * (d.P ? d.P : 1), 0));
    CK(dev_alloc(h, &d.inbox_cnt, N, 0));
    CK(dev_alloc(h, &d.inbox, (size_t)N * d.inbox_cap, 0));
    CK(dev_alloc(h, &d.first_suspect, NT, 0xFF));
    CK(dev_alloc(h, &d.crash_tick, NT, 0xFF));
    CK(dev_alloc(h, &d.g, (size_t)G_WORDS, 0));
    d.send_cnt = d.g + G_SEND;
    HK(hipHostMalloc(reinterpret_cast<void**>(&h->h_sync), G_WORDS * sizeof(uint32_t)));
    CK(dev_alloc(h, &d.ovf, (size_t)2 * d.ovf_cap, 0));
    CK(dev_alloc(h, &d.events, (size_t)d.event_cap, 0));
    CK(dev_alloc(h, &d.blk, ((size_t)d.nblocks + 1) * C_COUNT + 4096, 0));   // + the section-clock table of the measurement build
    CK(dev_alloc(h, &h->d_scratch64, (size_t)2, 0));
    HK(hipMemsetAsync(d.mb, (int)MB_UP, NT, h->stream));           // every member up, empty queue
  } else {
  d.ord_cap = 0; d.r_cap = d.p_cap = d.x_cap = 0;
  CK(dev_alloc(h, &d.minfo, NT, 0));
  CK(dev_alloc(h, &d.mb, ((size_t)NT + 3) & ~(size_t)3, 0));
  CK(dev_alloc(h, &d.probe_out, N, 0));
  CK(dev_alloc(h, &d.ackfrom, (size_t)N * (d.P ? d.P : 1), 0));
  CK(dev_alloc(h, &d.inbox_cnt, N, 0));
  CK(dev_alloc(h, &d.inbox, (size_t)N * d.inbox_cap, 0));
  CK(dev_alloc(h, &d.hot, N, 0));
  CK(dev_alloc(h, &d.todo, (size_t)d.todo_spill_at + d.todo_spill, 0));   // workgroup b reserves from region b mod TODO_REGIONS; the spill area behind the regions in use
  CK(dev_alloc(h, &d.todo_seg, N, 0));
  CK(dev_alloc(h, &d.todo_n, (size_t)(TODO_REGIONS + 1) * 16, 0));
  CK(dev_alloc(h, &d.kn_rec, N, 0));
  CK(dev_alloc(h, &d.pk, (size_t)N, 0));
  CK(dev_alloc(h, &d.inmask, (size_t)N, 0));
  CK(dev_alloc(h, &d.ackmask, (size_t)N, 0));
  CK(dev_alloc(h, &d.rum, (size_t)1 << RID_BITS, 0));
  CK(dev_alloc(h, &d.ring, KN_BITS, 0));
  CK(dev_alloc(h, &d.kw, N, 0));
  CK(dev_alloc(h, &d.kw_head, N, 0));
  CK(dev_alloc(h, &d.rtab, (size_t)d.R_phys * RT_WAYS, 0));
  CK(dev_alloc(h, &d.subject_of, (size_t)d.R_phys, 0));
  CK(dev_alloc(h, &d.fail, (size_t)N * (d.P ? d.P : 1), 0));
  CK(dev_alloc(h, &d.trow, (size_t)N * d.S, 0));
  // overflow cells per (deadline row, cycle parity): members that accept more than 7 suspicions in one tick --
  // a few per mille of the members without loss, one in three at 1 % loss and a million members
  // Sized for 288 GB: 4 cells per member and (row, parity), at most 24 GB in all.  Measured (profiles/r03w_*, r03ac_*): with
  // N / 2 cells at 30 % loss, or N / 16 with 25 crashes and 25 rejoins per tick among 2 M members (every member accepts ~25
  // suspicions per tick: four chained cells), the pools ran dry, the cells said "look at every view row" and 60-90 % of
  // merge_kernel's wave time went into those scans.  The pools are only touched when used.
  d.tovf_cap = (uint32_t)std::max<uint64_t>(1024u, std::min<uint64_t>(4ull * N, 24000000000ull / (32ull * d.S)));
  d.tovf_nsub = 1;
  while (d.tovf_nsub < 64u && d.tovf_nsub * 2u <= d.nblocks) d.tovf_nsub *= 2u;
  d.tovf_sub_cap = d.tovf_cap / d.tovf_nsub;
  d.tovf_cap = d.tovf_sub_cap * d.tovf_nsub;
  CK(dev_alloc(h, &d.tovf, (size_t)d.S * 2 * d.tovf_cap, 0));
  CK(dev_alloc(h, &d.tovf_n, (size_t)d.S * 2 * d.tovf_nsub * 16, 0));
  {
    const size_t cells = (size_t)(VTILE ? (N + VTILE - 1) / VTILE * VTILE : N) * d.R_phys;
#if SWIM_VSPLIT
    CK(dev_alloc(h, &d.Vk, cells, 0));
    CK(dev_alloc(h, &d.Vs, cells, 0));
#else
    CK(dev_alloc(h, &d.V, cells, 0));
#endif
  }
  CK(dev_alloc(h, &d.slot_last, (size_t)d.R_phys, 0xFF));
  CK(dev_alloc(h, &d.slot_base, (size_t)d.R_phys, 0));
  CK(dev_alloc(h, &d.slot_used, (size_t)d.R_phys, 0));
  CK(dev_alloc(h, &d.base_key, (size_t)NT, 0));
  CK(dev_alloc(h, &d.base_since, (size_t)NT, 0));
  CK(dev_alloc(h, &d.free_rows, (size_t)d.R_phys, 0));
  CK(dev_alloc(h, &d.settle_slots, (size_t)d.R_phys, 0));
  CK(dev_alloc(h, &d.settle_key, (size_t)d.R_phys, 0));
  CK(dev_alloc(h, &d.zero_slots, (size_t)d.R_phys, 0));
  CK(dev_alloc(h, &d.slot_born, (size_t)d.R_phys, 0));
  CK(dev_alloc(h, &d.settle_part, d.G ? (size_t)d.R_phys * d.nblocks : 1, 0));
  CK(dev_alloc(h, &d.pb, (size_t)2 * N * PB_SLOTS, 0));
  CK(dev_alloc(h, &d.first_suspect, NT, 0xFF));
  CK(dev_alloc(h, &d.crash_tick, NT, 0xFF));
  CK(dev_alloc(h, &d.g, (size_t)G_WORDS, 0));
  d.send_cnt = d.g + G_SEND;
  HK(hipHostMalloc(reinterpret_cast<void**>(&h->h_sync), G_WORDS * sizeof(uint32_t)));
  CK(dev_alloc(h, &d.ovf, (size_t)2 * d.ovf_cap, 0));
  CK(dev_alloc(h, &d.events, (size_t)d.event_cap, 0));
  CK(dev_alloc(h, &d.blk, ((size_t)d.nblocks + 1) * C_COUNT + 4096, 0));   // + the section-clock table of the measurement build
  CK(dev_alloc(h, &h->d_scratch64, (size_t)2, 0));
  if (d.n_shards > 1) {
    // exchange buffers (swim_device.h "cross-shard exchange"), sized from the expected traffic with headroom; overruns are
    // loud (SWIMSIM_ERR_CAPACITY), never silent drops
    const double l_ = c.loss_ppm / 1e6, pfail = 1.0 - (1.0 - l_) * (1.0 - l_);
    // records a probe block hands to the ingests: one per direct probe (a Ping's payload for a remote target; an Ack's that
    // needs more than a mask translation), and up to four hops per proxy of every direct probe that fails
    const double per_member = std::max(1u, d.P) * (2.0 + 4.0 * std::max(1u, d.K) * pfail);
    d.ord_cap = (uint32_t)std::min<double>(BLOCK * 2048.0, BLOCK * (1.5 * per_member + 6.0 * std::sqrt(per_member) + 4.0));
    // round 2: records per owner -- my members' deliveries spread over the shards; a degraded cluster under heavy loss sends
    // everything to the few members it still holds Alive (they may all sit on one shard): up to 65 536 members the hard bound
    const double per_peer = (double)N * per_member / d.n_shards;
    // ... and the segment a shard keeps for ITSELF takes every Ack of a remote target in a tick in which its masks are off (a
    // burst of rumour ids: a few ticks per hundred in the saturated regime): P per member, plus the chains' hops
    d.p_cap = (uint32_t)std::min<double>(4.0e8, std::max(std::max(per_peer * 2.0 + 8.0 * std::sqrt(per_peer) + 4096.0,
                                                                     (double)N * std::max(1u, d.P) * (1.0 + 4.0 * std::max(1u, d.K) * pfail) * 1.25 + 4096.0),
                                                            N <= 65536u ? (double)N * std::max(1u, d.P) * (2.0 + 4.0 * std::max(1u, d.K)) : 0.0));
    d.x_cap = 0;
    // round 1: the queues that travel as lists -- in a tick without masks every member's
    d.r_cap = XLINE_RECS * (N + 64u);
    CK(dev_alloc(h, &d.ord, (size_t)d.nblocks * d.ord_cap, 0));
    CK(dev_alloc(h, &d.r_send, (size_t)DICT_RECS + d.r_cap, 0));
    CK(dev_alloc(h, &d.r_recv, (size_t)d.n_shards * (DICT_RECS + d.r_cap), 0));
    CK(dev_alloc(h, &d.q_send, (size_t)d.n_shards * d.p_cap, 0));
    CK(dev_alloc(h, &d.q_recv, (size_t)d.n_shards * d.p_cap, 0));
    CK(dev_alloc(h, &d.xl, (size_t)d.n_shards * DICT_ENTRIES, 0xFF));
    CK(dev_alloc(h, &d.xidx, (size_t)NT, 0xFF));
    // foreign lines: normally a handful per tick (a rumour's first tick abroad), but in a tick without masks every
    // delivery from another shard becomes one
    d.fl_dyn_base = 0;
    d.fl_dyn_cap = (uint32_t)std::min<double>(4.0e8, (double)N * per_member * 1.5 + 65536.0);
    // (+ the foreign lines of injected rumours, swimsim_inject_rumor, behind the exchange's)
    d.fl_inj_base = d.fl_dyn_cap;
    CK(dev_alloc(h, &d.fl, ((size_t)d.fl_inj_base + INJECT_CAP) * 4, 0));
    CK(dev_alloc(h, &h->d_inject, (size_t)INJECT_CAP, 0));
    CK(dev_alloc(h, &d.mask_all, (size_t)NT, 0));
    CK(dev_alloc(h, &d.q_all, ((size_t)NT + 15) & ~(size_t)15, 0));
    if (d.G) {                                   // settling: every shard's word about its rows, all-gathered per tick
      d.s_cap = d.R_phys;
      CK(dev_alloc(h, &d.s_send, (size_t)d.n_shards * d.s_cap, 0));
      CK(dev_alloc(h, &d.s_recv, (size_t)d.n_shards * d.s_cap, 0));
      CK(dev_alloc(h, &d.settle_acc, (size_t)NT, 0));
    }
    if (d.join_pull || d.pull_T) {               // state pulls from hosts on other shards (round 0): a mass restart
      d.j_cap = std::max<uint32_t>(1u << 16, 16u * d.R_phys);   // of J members costs J x (entries a host holds) records
      // periodic pulls (pull_ticks = T): NT / T pullers per tick, a shard's share of their hosts sends what each host holds -- under
      // loss that is every row in use (3 000 members at 15 % loss, T = 3: 250 pulls x 1 500 entries to one peer per tick, the GPU
      // test that found the first sizing too small): room for min(rows, 2 048) entries per pull and twice the share, at most 8 M
      // records per peer (a loud capacity error beyond: SWIMSIM_ERR_CAPACITY)
      if (d.pull_T) d.j_cap = (uint32_t)std::min<double>(8388608.0, std::max<double>(d.j_cap, (d.push_pull ? 4.0 : 2.0) * std::min<double>(d.R_phys, 2048.0) * ((double)NT / d.pull_T / d.n_shards + 64.0)));   // (push_pull: the pullers' maps travel too)
      CK(dev_alloc(h, &d.j_send, (size_t)d.n_shards * d.j_cap, 0));
      CK(dev_alloc(h, &d.j_recv, (size_t)d.n_shards * d.j_cap, 0));
    }
  }
  if (d.n_shards == 1) {                         // the foreign lines of injected rumours (swimsim_inject_rumor)
    CK(dev_alloc(h, &d.fl, (size_t)INJECT_CAP * 4, 0));
    CK(dev_alloc(h, &h->d_inject, (size_t)INJECT_CAP, 0));
  }
  }
  CK(dev_alloc(h, &h->d_state, (size_t)1, 0));
  HK(hipMemcpyAsync(h->d_state, &d, sizeof d, hipMemcpyHostToDevice, h->stream));   // behind the allocation's memset
  if (!d.C) hipLaunchKernelGGL(init_members_kernel, dim3((NT + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, h->stream, d.minfo, d.mb, NT);
  HK(hipGetLastError());
  HK(hipStreamSynchronize(h->stream));
#undef CK
#undef HK
  *out = h;
  return SWIMSIM_OK;
}

int swimsim_create_msg(const swimsim_config_t* cfg, swimsim_t** out, char* err, size_t errcap) {
  const int rc = swimsim_create(cfg, out);
  if (err && errcap) { std::snprintf(err, errcap, "%s", rc ? g_create_err.c_str() : ""); }
  return rc;
}

int swimsim_inject_rumor(swimsim_t* h, uint32_t observer, uint32_t subject, uint8_t state, uint32_t incarnation) {
  if (!h) return SWIMSIM_ERR_INVALID;
  if (observer >= h->d.NT || subject >= h->d.NT || state > 2 || incarnation > INC_MAX)
    return set_err(h, SWIMSIM_ERR_INVALID, "inject_rumor: bad member / state / incarnation");
  if (observer - h->d.lo >= h->d.N) return set_err(h, SWIMSIM_ERR_INVALID, "inject_rumor: the observer is a member of another shard (its owner takes the message; the others: swimsim_note_outside_rumor)");
  if (h->d.C) return set_err(h, SWIMSIM_ERR_INVALID, "inject_rumor: not available with bounded member maps (view_cap)");
  if (h->injections.size() >= INJECT_CAP) return set_err(h, SWIMSIM_ERR_BUFFER, "inject_rumor: more than 4096 rumours before the next tick");
  h->injections.push_back(InjectRec{observer, subject, (incarnation << 2) | state, 0u});
  return SWIMSIM_OK;
}

int swimsim_note_outside_rumor(swimsim_t* h, uint32_t observer, uint32_t subject) {
  if (!h) return SWIMSIM_ERR_INVALID;
  if (observer >= h->d.NT || subject >= h->d.NT) return set_err(h, SWIMSIM_ERR_INVALID, "note_outside_rumor: bad member");
  if (h->d.C || h->d.n_shards < 2 || observer - h->d.lo < h->d.N) return set_err(h, SWIMSIM_ERR_INVALID, "note_outside_rumor: for the shards that do NOT own the observer of a message from outside");
  if (h->injections.size() >= INJECT_CAP) return set_err(h, SWIMSIM_ERR_BUFFER, "note_outside_rumor: more than 4096 before the next tick");
  h->injections.push_back(InjectRec{observer, subject, 0u, 1u});
  return SWIMSIM_OK;
}

int swimsim_get_config(const swimsim_t* h, swimsim_config_t* out) {
  if (!h || !out) return SWIMSIM_ERR_INVALID;
  *out = h->cfg;
  return SWIMSIM_OK;
}

int swimsim_schedule_fault(swimsim_t* h, uint64_t tick, uint32_t member, uint8_t up) {
  if (!h) return SWIMSIM_ERR_INVALID;
  if (member >= h->d.NT || up > 1) return set_err(h, SWIMSIM_ERR_INVALID, "schedule_fault: bad member/up");
  if (tick < h->tick || tick >= 0xFFFFFFFEull) return set_err(h, SWIMSIM_ERR_INVALID, "schedule_fault: tick in the past");
  // a sharded tick in progress has applied (phase 0) or uploaded (phase 1) its slice of the schedule: a change for that very
  // tick would shift the slice under it (an applied fault kept and applied again, the new one dropped)
  if (tick == h->tick && (h->begun || h->shard_phase != 0))
    return set_err(h, SWIMSIM_ERR_STATE, "schedule_fault: the current tick is in progress on this shard (schedule it before phase 0 / phase 1)");
  Fault f{(uint32_t)tick, member, up, h->fault_order++};
  auto pos = std::upper_bound(h->faults.begin(), h->faults.end(), f, [](const Fault& a, const Fault& b) {
    return a.tick != b.tick ? a.tick < b.tick : a.order < b.order; });
  h->faults.insert(pos, f);
  return SWIMSIM_OK;
}

// upload the fault records of the next nticks ticks; *fend = how many of h->faults they are
static int upload_faults(swimsim* h, uint32_t nticks, size_t* fend_out) {
  size_t fend = 0;
  while (fend < h->faults.size() && (uint64_t)h->faults[fend].tick < h->tick + nticks) ++fend;
  if (fend) {
    // within a tick the changes of one member stay in schedule order, members are grouped: begin_kernel
    // applies the members in parallel (changes of different members commute)
    std::vector<Fault> grouped(h->faults.begin(), h->faults.begin() + (long)fend);
    std::stable_sort(grouped.begin(), grouped.end(), [](const Fault& x, const Fault& y) {
      return x.tick != y.tick ? x.tick < y.tick : x.member < y.member; });
    std::vector<FaultRec> recs(fend);
    for (size_t x = 0; x < fend; ++x) recs[x] = FaultRec{grouped[x].member, grouped[x].up};
    if (fend > h->d_faults_cap) {
      HIPCHK(h, hipStreamSynchronize(h->stream));
      if (h->d_faults) HIPCHK(h, hipFree(h->d_faults));
      if (h->d_joined) HIPCHK(h, hipFree(h->d_joined));
      h->d_faults = nullptr; h->d_joined = nullptr; h->d_faults_cap = 0;
      const size_t cap = std::max<size_t>(1024, fend * 2);
      HIPCHK(h, hipMalloc(reinterpret_cast<void**>(&h->d_faults), cap * sizeof(FaultRec)));
      HIPCHK(h, hipMalloc(reinterpret_cast<void**>(&h->d_joined), cap * sizeof(uint32_t)));
      h->d_faults_cap = cap;
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));   // previous call's begin kernels are done with the buffer
    HIPCHK(h, hipMemcpy(h->d_faults, recs.data(), fend * sizeof(FaultRec), hipMemcpyHostToDevice));
  }
  *fend_out = fend;
  return SWIMSIM_OK;
}

int swimsim_step(swimsim_t* h, uint32_t nticks) {
  if (!h) return SWIMSIM_ERR_INVALID;
  if (h->poisoned) return set_err(h, SWIMSIM_ERR_STATE, "handle is poisoned by an earlier capacity error");
  if (h->d.n_shards > 1) return set_err(h, SWIMSIM_ERR_STATE, "a shard is stepped with swimsim_shard_phase1/2/3 (its peers must take part)");
  HIPCHK(h, hipSetDevice(h->device));
  size_t fend = 0;
  { int rc_ = upload_faults(h, nticks, &fend); if (rc_) return rc_; }
  if (h->timing) {
    while (h->ev_pool.size() < (size_t)nticks * 3) {
      hipEvent_t e;
      HIPCHK(h, hipEventCreate(&e));
      h->ev_pool.push_back(e);
    }
  }
  size_t fpos = 0;
  // measurement knob: SWIMSIM_GRAPH=1 captures the call's launches into ONE HIP graph and launches that --
  // what a graph does to the boundaries between the tick's kernels (scripts/graph_time.py)
  static const bool want_graph = [] { const char* e = std::getenv("SWIMSIM_GRAPH"); return e && e[0] == '1'; }();
  const bool graph = want_graph && !h->timing && h->injections.empty() && nticks > 0;
  if (graph) HIPCHK(h, hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
  // (an early return between here and hipStreamEndCapture must not leave the stream capturing: every later call would fail)
  struct CaptureGuard { swimsim* h; bool on; ~CaptureGuard() { if (on) { hipGraph_t g = nullptr; (void)hipStreamEndCapture(h->stream, &g); if (g) (void)hipGraphDestroy(g); } } } capture{h, graph};
  for (uint32_t k = 0; k < nticks; ++k) {
    const uint32_t t = (uint32_t)h->tick;
    hipEvent_t* ev = h->timing ? &h->ev_pool[(size_t)k * 3] : nullptr;
    const size_t f0 = fpos;
    while (fpos < fend && h->faults[fpos].tick <= t) ++fpos;
    const uint32_t tk = tick_key(h->cfg.seed, t);
    if (h->d.C) {                                   // bounded member maps: swim_sparse.h, one wave per member
      if (fpos > f0) hipLaunchKernelGGL(sp_begin_kernel, dim3(1), dim3(BLOCK), 0, h->stream, h->d, t, h->d_faults + f0, (uint32_t)(fpos - f0));
      launch_sparse_tick(h, t, tk, ev);
      h->tick++;
      continue;
    }
    uint32_t part = 3u;
    { bool inj = false; const int rc_ = flush_injections(h, t, &inj); if (rc_) return rc_; if (inj) part |= 4u; }   // begin_kernel: explicit records exist already
    uint32_t nup = 0;                               // upper bound of this tick's joins
    if (h->d.join_pull) for (size_t f = f0; f < fpos; ++f) nup += h->faults[f].up != 0;
    const uint32_t T = h->d.pull_T, first = T ? t % T : 0u;
    const uint32_t npp = (T && first < h->d.N) ? (h->d.N - first + T - 1u) / T : 0u;   // this tick's periodic pullers
    if (nup + npp) {
      // a tick with state pulls: between the two parts of the start of the tick, one block per puller
      hipLaunchKernelGGL(begin_kernel, dim3(1), dim3(BLOCK), 0, h->stream, h->d, t, tk, h->d_faults + f0, (uint32_t)(fpos - f0),
                         h->d_joined, 1u, PeerCounts{}, JoinView{});
      hipLaunchKernelGGL(join_pull_kernel, dim3(std::min(nup + npp, 16384u)), dim3(BLOCK), 0, h->stream, h->d, t, tk, h->d_faults + f0,
                         (uint32_t)(fpos - f0), h->d_joined, nup);
      // push-pull: once every pull has read its host, the hosts merge their pullers' maps
      if (h->d.push_pull && npp) hipLaunchKernelGGL(push_kernel, dim3(std::min(npp, 16384u)), dim3(BLOCK), 0, h->stream, h->d, t, tk, h->d_faults + f0, (uint32_t)(fpos - f0));
      part = (part & ~1u) | 8u;
    }
    // A PLAIN tick -- no scheduled change, no message from outside, no state pull, no settling, no explicit-record kernel -- needs
    // nothing of begin_kernel but the window heads, the tick's ring and three resets: probe_kernel's workgroup 0 does that on the
    // side (`fold`), the launch and its kernel boundary are saved (SWIMSIM_FOLD_BEGIN=0 at create: always launch it; A/B, tests)
    // ... and so does a tick whose scheduled changes are a few crashes of different members (round 5: the benchmarked regime has one
    // per tick): the list rides into probe_kernel as an overlay on ground truth (swim_kernels.h CrashList)
    CrashList cl{};
    bool crashes_only = fpos - f0 <= FOLD_MAX_CRASHES;
    for (size_t f = f0; f < fpos && crashes_only; ++f) {
      crashes_only = h->faults[f].up == 0;
      for (size_t g2 = f0; g2 < f; ++g2) crashes_only = crashes_only && h->faults[g2].member != h->faults[f].member;
      if (crashes_only) cl.member[cl.n++] = h->faults[f].member;
    }
    if (!crashes_only) cl = CrashList{};
    const uint32_t fold = (h->fold_begin && part == 3u && crashes_only && !(nup + npp) && !h->d.G && !h->d.strict && !records_kernel_every_tick(h)) ? 1u : 0u;   // (strict rules: begin_kernel declares the ids untrusted in every tick)
    if (!fold) cl = CrashList{};
    if (!fold)
      hipLaunchKernelGGL(begin_kernel, dim3(1), dim3(BLOCK), 0, h->stream, h->d, t, tk, h->d_faults + f0, (uint32_t)(fpos - f0),
                         h->d_joined, part, PeerCounts{}, JoinView{});
    launch_tick(h, t, tk, ev, fold, cl);
    h->tick++;
  }
  h->faults.erase(h->faults.begin(), h->faults.begin() + (long)fpos);
  // the last tick's settling is committed before anybody reads state (between ticks begin_kernel does it)
  if (h->d.G && nticks) hipLaunchKernelGGL(settle_flush_kernel, dim3(1), dim3(BLOCK), 0, h->stream, h->d);
  if (graph) {
    hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
    capture.on = false;
    HIPCHK(h, hipStreamEndCapture(h->stream, &g));
    HIPCHK(h, hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    const auto w0 = std::chrono::steady_clock::now();
    HIPCHK(h, hipGraphLaunch(ge, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->graph_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count();
    (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
  }
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (h->timing) {
    for (uint32_t k = 0; k < nticks; ++k) {
      float a = 0, b = 0;
      HIPCHK(h, hipEventElapsedTime(&a, h->ev_pool[(size_t)k * 3], h->ev_pool[(size_t)k * 3 + 1]));
      HIPCHK(h, hipEventElapsedTime(&b, h->ev_pool[(size_t)k * 3 + 1], h->ev_pool[(size_t)k * 3 + 2]));
      h->probe_ms += a; h->merge_ms += b;
    }
    h->timed_ticks += nticks;
  }
  return check_device_errors(h);
}

int swimsim_tick(const swimsim_t* h, uint64_t* tick) {
  if (!h || !tick) return SWIMSIM_ERR_INVALID;
  *tick = h->tick;
  return SWIMSIM_OK;
}

int swimsim_drain_events(swimsim_t* h, swimsim_event_t* buf, size_t cap, size_t* n_out) {
  if (!h || !n_out) return SWIMSIM_ERR_INVALID;
  HIPCHK(h, hipSetDevice(h->device));
  int rc = pull_events(h);
  if (rc) return rc;
  const size_t n = h->host_events.size();
  *n_out = n;
  if (n > cap || (n && !buf)) return SWIMSIM_ERR_BUFFER;
  if (n) std::memcpy(buf, h->host_events.data(), n * sizeof *buf);
  h->host_events.clear();
  return SWIMSIM_OK;
}

// one observer's cells of every view row in use (reclaimed rows read as empty) and the rows' subjects
static int read_column(swimsim_t* h, uint32_t observer, std::vector<uint2>* col, std::vector<uint32_t>* subj) {
  uint32_t g[G_WORDS];
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipMemcpy(g, h->d.g, sizeof g, hipMemcpyDeviceToHost));
  const uint32_t ns = std::min(g[G_NSLOTS], h->d.R_phys);
  col->resize(ns); subj->resize(ns);
  if (!ns) return SWIMSIM_OK;
  // one observer's entries are a strided column of V (stride = one row of its tile)
#if SWIM_VSPLIT
  {
    // the two planes of the column (swim_device.h: key << 8 | low byte of lastChange + 1; lastChange + 1)
    std::vector<uint32_t> wk(ns), ws(ns);
    const size_t at = vidx_of(h->d.N, h->d.R_phys, observer, 0), pitch = (VTILE ? (size_t)VTILE : (size_t)h->d.N) * sizeof(uint32_t);
    HIPCHK(h, hipMemcpy2D(wk.data(), sizeof(uint32_t), h->d.Vk + at, pitch, sizeof(uint32_t), ns, hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemcpy2D(ws.data(), sizeof(uint32_t), h->d.Vs + at, pitch, sizeof(uint32_t), ns, hipMemcpyDeviceToHost));
    for (uint32_t r = 0; r < ns; ++r) (*col)[r] = make_uint2(wk[r] >> 8, ws[r]);
  }
#else
  HIPCHK(h, hipMemcpy2D(col->data(), sizeof(uint2), h->d.V + vidx_of(h->d.N, h->d.R_phys, observer, 0),
                        (VTILE ? (size_t)VTILE : (size_t)h->d.N) * sizeof(uint2), sizeof(uint2), ns, hipMemcpyDeviceToHost));
#endif
  HIPCHK(h, hipMemcpy(subj->data(), h->d.subject_of, (size_t)ns * sizeof(uint32_t), hipMemcpyDeviceToHost));
  std::vector<uint8_t> used(ns);
  HIPCHK(h, hipMemcpy(used.data(), h->d.slot_used, ns, hipMemcpyDeviceToHost));
  for (uint32_t r = 0; r < ns; ++r)
    if (!used[r]) { (*col)[r] = make_uint2(0u, 0u); (*subj)[r] = NONE32; }
  return SWIMSIM_OK;
}

// bounded member maps: one member's map as stored, rows[0..C) subjects, [C..2C) keys, [2C..3C) lastChange + 1; *n entries in use
static int read_sparse_map(swimsim_t* h, uint32_t member, std::vector<uint32_t>* rows, uint32_t* n) {
  HIPCHK(h, hipStreamSynchronize(h->stream));
  rows->resize((size_t)3 * h->d.C);
  const uint32_t ml = member - h->d.lo;            // (the caller has checked that this handle owns the member)
  HIPCHK(h, hipMemcpy(rows->data(), h->d.sp_tab + (size_t)ml * 3 * h->d.C, rows->size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
  HIPCHK(h, hipMemcpy(n, h->d.sp_tab_n + ml, sizeof(uint32_t), hipMemcpyDeviceToHost));
  if (*n > h->d.C) *n = h->d.C;
  return SWIMSIM_OK;
}

int swimsim_read_view(swimsim_t* h, uint32_t observer, swimsim_view_entry_t* buf, size_t cap, size_t* n_out) {
  if (!h || !n_out || observer - h->d.lo >= h->d.N) return SWIMSIM_ERR_INVALID;   // not a member this handle owns
  HIPCHK(h, hipSetDevice(h->device));
  if (h->d.C) {                                     // bounded map: its entries, sorted by subject
    std::vector<uint32_t> rows;
    uint32_t n = 0;
    int rc0 = read_sparse_map(h, observer, &rows, &n);
    if (rc0) return rc0;
    std::vector<swimsim_view_entry_t> es(n);
    for (uint32_t e = 0; e < n; ++e) {
      es[e] = swimsim_view_entry_t{};
      es[e].subject = rows[e]; es[e].incarnation = rows[h->d.C + e] >> 2; es[e].state = (uint8_t)(rows[h->d.C + e] & 3u);
      es[e].since_tick = rows[2 * h->d.C + e] - 1;
    }
    std::sort(es.begin(), es.end(), [](const swimsim_view_entry_t& a, const swimsim_view_entry_t& b) { return a.subject < b.subject; });
    *n_out = n;
    if (n > cap || (n && !buf)) return SWIMSIM_ERR_BUFFER;
    if (n) std::memcpy(buf, es.data(), n * sizeof *buf);
    return SWIMSIM_OK;
  }
  std::vector<uint2> col; std::vector<uint32_t> subj;
  int rc = read_column(h, observer - h->d.lo, &col, &subj);
  if (rc) return rc;
  std::vector<swimsim_view_entry_t> ents;
  std::vector<uint32_t> own;                       // subjects listed with the observer's own entry
  for (size_t r = 0; r < col.size(); ++r) {
    if (col[r].x == 0 || subj[r] == observer) continue;
    swimsim_view_entry_t e{};
    e.subject = subj[r]; e.incarnation = col[r].x >> 2; e.state = (uint8_t)(col[r].x & 3u); e.since_tick = col[r].y - 1;
    ents.push_back(e);
    own.push_back(subj[r]);
  }
  if (h->d.G) {
    // settled subjects: Alive@i (i > 0) stays a listed member, Dead ones were removed (removeDeadNodes).  The list of
    // settled-Alive subjects is the same for every observer: fetched once per tick (two N-sized copies and a scan per
    // CALL made reading many observers' views quadratic)
    if (h->settled_alive_tick != h->tick) {
      std::vector<uint32_t> bk(h->d.NT), bs(h->d.NT);
      HIPCHK(h, hipMemcpy(bk.data(), h->d.base_key, (size_t)h->d.NT * 4, hipMemcpyDeviceToHost));
      HIPCHK(h, hipMemcpy(bs.data(), h->d.base_since, (size_t)h->d.NT * 4, hipMemcpyDeviceToHost));
      h->settled_alive.clear();
      for (uint32_t sj = 0; sj < h->d.NT; ++sj)
        if (bk[sj] && (bk[sj] & 3u) == ST_ALIVE) {
          swimsim_view_entry_t e{};
          e.subject = sj; e.incarnation = bk[sj] >> 2; e.state = ST_ALIVE; e.since_tick = bs[sj];
          h->settled_alive.push_back(e);
        }
      h->settled_alive_tick = h->tick;
    }
    std::sort(own.begin(), own.end());
    for (const swimsim_view_entry_t& e : h->settled_alive)
      if (e.subject != observer && !std::binary_search(own.begin(), own.end(), e.subject)) ents.push_back(e);
  }
  std::sort(ents.begin(), ents.end(), [](const swimsim_view_entry_t& a, const swimsim_view_entry_t& b) { return a.subject < b.subject; });
  *n_out = ents.size();
  if (ents.size() > cap || (!ents.empty() && !buf)) return SWIMSIM_ERR_BUFFER;
  if (!ents.empty()) std::memcpy(buf, ents.data(), ents.size() * sizeof *buf);
  return SWIMSIM_OK;
}

int swimsim_read_member(swimsim_t* h, uint32_t m, swimsim_member_t* out) {
  if (!h || !out || m - h->d.lo >= h->d.N) return SWIMSIM_ERR_INVALID;           // not a member this handle owns
  HIPCHK(h, hipSetDevice(h->device));
  const uint32_t ml = m - h->d.lo;
  if (h->d.C) {
    std::vector<uint32_t> rows;
    uint32_t n = 0;
    int rc0 = read_sparse_map(h, m, &rows, &n);
    if (rc0) return rc0;
    uint2 hot; uint8_t b;
    HIPCHK(h, hipMemcpy(&hot, h->d.hot + ml, sizeof hot, hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemcpy(&b, h->d.mb + m, 1, hipMemcpyDeviceToHost));
    std::memset(out, 0, sizeof *out);
    out->id = m; out->incarnation = hot.x; out->up = b & MB_UP;
    const uint32_t qn = (b >> MB_PBN_SHIFT) & 0xFu;
    if (qn) {
      uint2 line[PB_SLOTS];
      HIPCHK(h, hipMemcpy(line, h->d.sp_q + ((size_t)(h->tick & 1u) * h->d.N + ml) * PB_SLOTS, sizeof line, hipMemcpyDeviceToHost));
      for (uint32_t q = 0; q < qn && q < (uint32_t)PB_SLOTS; ++q) {
        swimsim_rumor_t& r = out->rumors[out->n_rumors++];
        r.subject = line[q].x; r.incarnation = pe_key(line[q].y) >> 2; r.state = (uint8_t)(pe_key(line[q].y) & 3u); r.tx_left = (uint8_t)pe_tx(line[q].y);
      }
    }
    uint32_t nt = 0;
    for (uint32_t e = 0; e < n; ++e) nt += (rows[h->d.C + e] & 3u) == ST_SUSPECT;
    out->n_timers = (uint16_t)nt;
    return SWIMSIM_OK;
  }
  std::vector<uint2> col; std::vector<uint32_t> subj;
  int rc = read_column(h, ml, &col, &subj);
  if (rc) return rc;
  uint2 hot; uint32_t mi;
  HIPCHK(h, hipMemcpy(&hot, h->d.hot + ml, sizeof hot, hipMemcpyDeviceToHost));
  HIPCHK(h, hipMemcpy(&mi, h->d.minfo + m, sizeof mi, hipMemcpyDeviceToHost));
  std::memset(out, 0, sizeof *out);
  out->id = m; out->incarnation = hot.x; out->up = (mi >> 21) & 1u;
  if ((mi >> 16) & 15u) {
    uint64_t line[PB_SLOTS];
    HIPCHK(h, hipMemcpy(line, h->d.pb + ((size_t)((mi >> 20) & 1u) * h->d.N + ml) * PB_SLOTS, sizeof line, hipMemcpyDeviceToHost));
    for (int q = 0; q < PB_SLOTS; ++q) {
      const uint32_t lo = (uint32_t)line[q], hi = (uint32_t)(line[q] >> 32);
      if (!pe_tx(hi)) continue;
      swimsim_rumor_t& r = out->rumors[out->n_rumors++];
      const uint32_t sl = pe_slot(lo);
      r.subject = sl < subj.size() ? subj[sl] : NONE32;
      r.incarnation = pe_key(hi) >> 2; r.state = (uint8_t)(pe_key(hi) & 3u); r.tx_left = (uint8_t)pe_tx(hi);
    }
  }
  uint32_t nt = 0;
  for (size_t r = 0; r < col.size(); ++r) if (subj[r] != m && col[r].x && (col[r].x & 3u) == ST_SUSPECT) nt++;
  out->n_timers = (uint16_t)nt;
  return SWIMSIM_OK;
}

int swimsim_first_detect(swimsim_t* h, uint64_t* out, size_t n) {
  if (!h || !out || n != h->d.NT) return SWIMSIM_ERR_INVALID;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  std::vector<uint32_t> tmp(n);
  HIPCHK(h, hipMemcpy(tmp.data(), h->d.first_suspect, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
  for (size_t j = 0; j < n; ++j) out[j] = tmp[j] == NONE32 ? SWIMSIM_TICK_NONE : tmp[j];
  return SWIMSIM_OK;
}

int swimsim_digest(swimsim_t* h, uint64_t* out) {
  if (!h || !out) return SWIMSIM_ERR_INVALID;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipMemsetAsync(h->d_scratch64, 0, sizeof(unsigned long long), h->stream));
  if (h->d.C) hipLaunchKernelGGL(sp_digest_kernel, dim3((h->d.N + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, h->stream, h->d, (uint32_t)h->tick, h->d_scratch64);
  else hipLaunchKernelGGL(digest_kernel, dim3(h->d.nblocks), dim3(BLOCK), 0, h->stream, h->d, h->d_scratch64);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipStreamSynchronize(h->stream));
  unsigned long long acc = 0;
  HIPCHK(h, hipMemcpy(&acc, h->d_scratch64, sizeof acc, hipMemcpyDeviceToHost));
  // a shard returns its members' part; shard 0 adds the tick term, so the parts simply add up
  *out = (h->d.shard == 0 ? mix64((uint64_t)TAG_TICK + h->tick) : 0ull) + acc;
  return SWIMSIM_OK;
}

int swimsim_coverage(swimsim_t* h, uint32_t subject, uint8_t state, uint32_t incarnation, uint64_t out[2]) {
  if (!h || !out || subject >= h->d.NT || state > SWIMSIM_DEAD || incarnation > INC_MAX) return SWIMSIM_ERR_INVALID;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipMemsetAsync(h->d_scratch64, 0, 2 * sizeof(unsigned long long), h->stream));
  if (h->d.C) hipLaunchKernelGGL(sp_coverage_kernel, dim3((h->d.N + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, h->stream, h->d, subject, (incarnation << 2) | state,
                                 h->d_scratch64);
  else hipLaunchKernelGGL(coverage_kernel, dim3(h->d.nblocks), dim3(BLOCK), 0, h->stream, h->d, subject, (incarnation << 2) | state,
                     h->d_scratch64);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipStreamSynchronize(h->stream));
  unsigned long long acc[2] = {0, 0};
  HIPCHK(h, hipMemcpy(acc, h->d_scratch64, sizeof acc, hipMemcpyDeviceToHost));
  out[0] = acc[0]; out[1] = acc[1];
  return SWIMSIM_OK;
}

int swimsim_counters(swimsim_t* h, uint64_t* out, size_t n) {
  if (!h || !out || n < SWIMSIM_CTR_COUNT) return SWIMSIM_ERR_INVALID;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  const size_t rows = (size_t)h->d.nblocks + 1;
  std::vector<uint64_t> blk(rows * C_COUNT);
  HIPCHK(h, hipMemcpy(blk.data(), h->d.blk, blk.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
  for (int c = 0; c < C_COUNT; ++c) out[c] = 0;
  for (size_t r = 0; r < rows; ++r)
    for (int c = 0; c < C_COUNT; ++c) out[c] += blk[r * C_COUNT + c];
  return SWIMSIM_OK;
}

int swimsim_table_stats(swimsim_t* h, uint64_t* out, size_t n) {
  if (!h || !out || n < 5) return SWIMSIM_ERR_INVALID;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  uint32_t g[G_WORDS];
  HIPCHK(h, hipMemcpy(g, h->d.g, sizeof g, hipMemcpyDeviceToHost));
  out[0] = g[G_NSLOTS]; out[1] = g[G_NLIVE]; out[2] = g[G_NFREE]; out[3] = g[G_NRUM]; out[4] = h->d.R_phys;   // (bounded member maps: all zero, no view rows)
  if (n >= 7) { out[5] = std::max(g[G_OVF0], g[G_OVF1]); out[6] = h->d.ovf_cap; }
  if (n >= 9 && h->d.C) { out[7] = h->sp_grid_probe; out[8] = h->sp_grid_merge; }   // bounded member maps: the persistent grids (workgroups)
  if (n >= 10) out[9] = (uint64_t)(h->graph_ms * 1000.0);                           // SWIMSIM_GRAPH=1: the last step's graph, launch to completion, us
#ifdef SWIM_REC_STATS
  if (n >= 10) { out[7] = g[90]; out[8] = g[91]; out[9] = g[92]; }
#endif   // inbox overflow list: entries in the fuller of the two, room
  return SWIMSIM_OK;
}

#ifdef SWIM_ABLATE
// measurement build only (scripts/ablate.py): which memory operations the tick kernels leave out from now on
int swimsim_debug_ablate(swimsim_t* h, uint32_t mask) { h->d.dbg = mask; return SWIMSIM_OK; }
#endif

#ifdef SWIM_SECTION_CLOCKS
// measurement build only (scripts/section_clocks.py): the 64 section-clock words (summed over their 64 copies), then zeroed
int swimsim_debug_sections(swimsim_t* h, uint64_t* out) {
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  uint64_t* p = h->d.blk + ((size_t)h->d.nblocks + 1) * C_COUNT;
  std::vector<uint64_t> tab(4096);
  HIPCHK(h, hipMemcpy(tab.data(), p, tab.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
  HIPCHK(h, hipMemset(p, 0, tab.size() * sizeof(uint64_t)));
  for (int k = 0; k < 64; ++k) { out[k] = 0; for (int c = 0; c < 64; ++c) out[k] += tab[(size_t)c * 64 + k]; }
  return SWIMSIM_OK;
}
#endif

int swimsim_k_random_members(swimsim_t* h, uint32_t observer, uint32_t n, const uint32_t* excludes,
                             size_t n_excludes, uint32_t* out, size_t cap, size_t* n_out) {
  if (!h || !n_out || observer - h->d.lo >= h->d.N || n > 255 || (n_excludes && !excludes)) return SWIMSIM_ERR_INVALID;
  if (h->d.C) return set_err(h, SWIMSIM_ERR_INVALID, "k_random_members: not available with bounded member maps (view_cap)");
  HIPCHK(h, hipSetDevice(h->device));
  const size_t need = 257 + n_excludes;
  if (need > h->d_sel_cap) {
    if (h->d_sel) HIPCHK(h, hipFree(h->d_sel));
    h->d_sel = nullptr; h->d_sel_cap = 0;
    HIPCHK(h, hipMalloc(reinterpret_cast<void**>(&h->d_sel), need * sizeof(uint32_t)));
    h->d_sel_cap = need;
  }
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (n_excludes) HIPCHK(h, hipMemcpy(h->d_sel + 257, excludes, n_excludes * sizeof(uint32_t), hipMemcpyHostToDevice));
  const uint32_t tk = tick_key(h->cfg.seed, (uint32_t)h->tick);
  hipLaunchKernelGGL(select_debug_kernel, dim3(1), dim3(64), 0, h->stream, h->d, tk, observer, n,
                     h->d_sel + 257, (uint32_t)n_excludes, h->d_sel, h->d_sel + 256);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipStreamSynchronize(h->stream));
  uint32_t res[257];
  HIPCHK(h, hipMemcpy(res, h->d_sel, sizeof res, hipMemcpyDeviceToHost));
  *n_out = res[256];
  if (res[256] > cap || (res[256] && !out)) return SWIMSIM_ERR_BUFFER;
  std::memcpy(out, res, res[256] * sizeof(uint32_t));
  return SWIMSIM_OK;
}

/* ---- sharded stepping: one tick = phase1 -> [exchange requests] -> phase2 -> [exchange payloads] -> phase3 ---- */

int swimsim_shard_info(const swimsim_t* h, uint32_t* lo, uint32_t* n_local, uint32_t* r_cap, uint32_t* p_cap, uint32_t* x_cap) {
  if (!h) return SWIMSIM_ERR_INVALID;
  if (lo) *lo = h->d.lo;
  if (n_local) *n_local = h->d.N;
  if (r_cap) *r_cap = h->d.C ? 0u : DICT_RECS + h->d.r_cap;      // the tick's dictionary, then the queues that travel as lists (bounded handles: no such records)
  if (p_cap) *p_cap = h->d.p_cap;
  if (x_cap) *x_cap = h->d.x_cap;
  return SWIMSIM_OK;
}

int swimsim_shard_buffers(swimsim_t* h, void** send /*[3]*/, void** recv /*[3]*/) {
  if (!h || !send || !recv) return SWIMSIM_ERR_INVALID;
  // dense handles: kind 0 = ONE segment for every peer (dictionary + lists), kind 1 = 8-byte records {dst, src}; bounded
  // handles: kind 1 = 16-byte records {dst, src, -, -}; kind 2: nothing any more
  send[0] = h->d.r_send; send[1] = h->d.C ? (void*)h->d.p_send : (void*)h->d.q_send; send[2] = nullptr;
  recv[0] = h->d.r_recv; recv[1] = h->d.C ? (void*)h->d.p_recv : (void*)h->d.q_recv; recv[2] = nullptr;
  return SWIMSIM_OK;
}

static int shard_check(swimsim* h, int phase) {
  if (!h) return SWIMSIM_ERR_INVALID;
  if (h->poisoned) return set_err(h, SWIMSIM_ERR_STATE, "handle is poisoned by an earlier capacity error");
  if (h->d.n_shards < 2) return set_err(h, SWIMSIM_ERR_STATE, "not a sharded handle (use swimsim_step)");
  if (h->shard_phase != phase) return set_err(h, SWIMSIM_ERR_STATE, "shard phases must run in the order 1, 2, 3");
  return SWIMSIM_OK;
}

// End of a phase: ONE small pinned copy brings the capacity flags and the per-peer send counts to the
// host, one stream synchronisation.  counts[k * n_shards + g], k = 0 round-1 records (dictionary + lists: the
// same segment for every peer), 1 round-2 records {dst, src}, 2 nothing any more.
static int finish_phase(swimsim* h, uint32_t* counts) {
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipMemcpyAsync(h->h_sync, h->d.g, G_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  const uint32_t* g = h->h_sync;
  if (g[G_ERR]) return check_device_errors(h);
  if (counts) {
    const uint32_t G = h->d.n_shards;
    const uint32_t* c = g + G_SEND;
    for (uint32_t p = 0; p < G; ++p) {
      counts[p] = p == h->d.shard ? 0u : DICT_RECS + std::min(g[G_XLINES] * XLINE_RECS, h->d.r_cap);   // the same segment for every peer
      counts[G + p] = p == h->d.shard ? 0u : std::min(c[MAX_SHARDS + p], h->d.p_cap);
      counts[2 * G + p] = 0u;
    }
  }
  return SWIMSIM_OK;
}

static PeerCounts peer_counts(const swimsim* h, const uint32_t* in) {
  PeerCounts pc{};
  for (uint32_t p = 0; p < h->d.n_shards; ++p) pc.v[p] = in[p];
  return pc;
}

/* join_pull on a sharded cluster: the start of the tick in two parts around exchange round 0 (kind 4).  phase0
 * applies the tick's faults; if members come up in this tick (every shard sees the same schedule, so every shard
 * gets the same answer) *round_needed = 1, the owners of their join hosts have written what the hosts know as
 * 16-byte records into send[p][0 .. counts[p]) of swimsim_shard_join_buffers, the caller delivers them like any
 * other round and reports the arrivals with swimsim_shard_join_ingest; then phase1.  Without joins in the tick
 * phase0 does nothing.  Optional (a no-op) when join_pull is off. */
int swimsim_shard_phase0(swimsim_t* h, uint32_t* counts, int* round_needed) {
  int rc = shard_check(h, 0);
  if (rc) return rc;
  if (!counts || !round_needed) return SWIMSIM_ERR_INVALID;
  const uint32_t G = h->d.n_shards;
  for (uint32_t p = 0; p < G; ++p) counts[p] = 0;
  *round_needed = 0;
  if ((!h->d.join_pull && !h->d.pull_T) || h->begun) return SWIMSIM_OK;
  bool joins = false;
  if (h->d.join_pull) for (const Fault& f : h->faults) { if (f.tick > h->tick) break; joins |= f.up != 0; }
  // periodic pulls (pull_ticks = T): the members t mod T, t mod T + T, ... of the whole population pull in this tick -- every tick has some
  const bool pulls = h->d.pull_T != 0 && (uint32_t)(h->tick % h->d.pull_T) < h->d.NT;
  if (!joins && !pulls) return SWIMSIM_OK;
  HIPCHK(h, hipSetDevice(h->device));
  size_t fend = 0;
  rc = upload_faults(h, 1, &fend);
  if (rc) return rc;
  const uint32_t t = (uint32_t)h->tick;
  const uint32_t tk = tick_key(h->cfg.seed, t);
  { bool inj = false; rc = flush_injections(h, t, &inj); if (rc) return rc; h->tick_inj = inj; }   // before the tick's scheduled changes (swimsim_step does the same)
  hipLaunchKernelGGL(begin_kernel, dim3(1), dim3(BLOCK), 0, h->stream, h->d, t, tk, h->d_faults, (uint32_t)fend, h->d_joined, 1u, PeerCounts{}, JoinView{});
  {
    const uint32_t T = h->d.pull_T, first = T ? t % T : 0u;
    const uint32_t items = (uint32_t)fend + ((T && first < h->d.NT) ? (h->d.NT - first + T - 1u) / T : 0u);   // joiners (at most) + the tick's periodic pullers, everywhere
    hipLaunchKernelGGL(pull_send_kernel, dim3(std::max(1u, std::min<uint32_t>(1024u, (items + BLOCK - 1) / BLOCK))), dim3(BLOCK), 0, h->stream,
                       h->d, t, tk, h->d_faults, (uint32_t)fend, h->d_joined);
  }
  rc = finish_phase(h, nullptr);
  if (rc) return rc;
  for (uint32_t p = 0; p < G; ++p) counts[p] = p == h->d.shard ? 0u : std::min(h->h_sync[G_JSEND + p], h->d.j_cap);
  h->begun = true; h->begun_fend = fend;
  *round_needed = 1;
  return SWIMSIM_OK;
}

int swimsim_shard_join_buffers(swimsim_t* h, void** send, void** recv, uint32_t* cap) {
  if (!h || !send || !recv) return SWIMSIM_ERR_INVALID;
  *send = h->d.j_send; *recv = h->d.j_recv;
  if (cap) *cap = h->d.j_cap;
  return SWIMSIM_OK;
}

int swimsim_shard_join_ingest(swimsim_t* h, const uint32_t* counts_in) {
  int rc = shard_check(h, 0);
  if (rc) return rc;
  if (!counts_in || !h->begun) return set_err(h, SWIMSIM_ERR_STATE, "join_ingest follows a swimsim_shard_phase0 that asked for round 0");
  for (uint32_t p = 0; p < h->d.n_shards; ++p) h->j_in[p] = counts_in[p];
  return SWIMSIM_OK;
}

int swimsim_shard_phase1(swimsim_t* h, uint32_t* counts) {
  int rc = shard_check(h, 0);
  if (rc) return rc;
  if (!counts) return SWIMSIM_ERR_INVALID;
  HIPCHK(h, hipSetDevice(h->device));
  if (h->d.C) {
    // a bounded shard (swim_sparse.h): the tick's scheduled changes, then its slice of the queue lines into the replica; round 1
    // all-gathers the lines and the member bytes (no records of kind 0)
    size_t fe = 0;
    rc = upload_faults(h, 1, &fe);
    if (rc) return rc;
    const uint32_t t = (uint32_t)h->tick;
    if (fe) hipLaunchKernelGGL(sp_begin_kernel, dim3(1), dim3(BLOCK), 0, h->stream, h->d, t, h->d_faults, (uint32_t)fe);
    h->faults.erase(h->faults.begin(), h->faults.begin() + (long)fe);
    hipLaunchKernelGGL(sp_publish_kernel, dim3((h->d.N * PB_SLOTS + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, h->stream, h->d, t);
    rc = finish_phase(h, nullptr);
    if (rc) return rc;
    for (uint32_t k = 0; k < 3 * h->d.n_shards; ++k) counts[k] = 0;
    h->shard_phase = 1;
    return SWIMSIM_OK;
  }
  size_t fend = h->begun_fend;
  if (!h->begun) { rc = upload_faults(h, 1, &fend); if (rc) return rc; }
  if (h->timing && !h->tick_ev[0]) for (int k = 0; k < 3; ++k) HIPCHK(h, hipEventCreate(&h->tick_ev[k]));
  const uint32_t t = (uint32_t)h->tick;
  const uint32_t tk = tick_key(h->cfg.seed, t);
  // one launch does the whole start of the tick, unless swimsim_shard_phase0 ran its first part already (join-time
  // pulls to exchange in between)
  if (!h->begun) { bool inj = false; rc = flush_injections(h, t, &inj); if (rc) return rc; h->tick_inj = inj; }
  if (h->d.pull_T && !h->begun) return set_err(h, SWIMSIM_ERR_STATE, "a shard with pull_ticks starts every tick with swimsim_shard_phase0 (its periodic pulls are exchange round 0)");
  if (h->begun) {                                   // the pulls from hosts on this shard: a block per puller (the ones the peers
    uint32_t nup = 0;                               // sent are merged by begin_kernel: other pullers, rows no local host holds)
    if (h->d.join_pull) for (size_t f = 0; f < fend; ++f) nup += h->faults[f].up != 0;
    const uint32_t T = h->d.pull_T, first = T ? (t % T + T - h->d.lo % T) % T : 0u;       // my first periodic puller, as a local index
    const uint32_t npp = (T && first < h->d.N) ? (h->d.N - first + T - 1u) / T : 0u;
    if (nup + npp) hipLaunchKernelGGL(join_pull_kernel, dim3(std::min(nup + npp, 4096u)), dim3(BLOCK), 0, h->stream, h->d, t, tk, h->d_faults,
                                      (uint32_t)fend, h->d_joined, nup);
    // push-pull: once every local pull has read its host, my hosts merge the maps of their pullers on THIS shard (those of pullers
    // elsewhere arrived as records: begin_kernel below)
    if (h->d.push_pull && npp) hipLaunchKernelGGL(push_kernel, dim3(std::min(npp, 4096u)), dim3(BLOCK), 0, h->stream, h->d, t, tk, h->d_faults, (uint32_t)fend);
  }
  hipLaunchKernelGGL(begin_kernel, dim3(1), dim3(BLOCK), 0, h->stream, h->d, t, tk, h->d_faults, (uint32_t)fend,
                     h->d_joined, (h->begun ? (2u | 8u) : 3u) | (h->tick_inj ? 4u : 0u), peer_counts(h, h->j_in), JoinView{});
  h->begun = false; h->tick_inj = false;
  std::fill(h->j_in, h->j_in + MAX_SHARDS, 0u);
  h->faults.erase(h->faults.begin(), h->faults.begin() + (long)fend);
  // my slice of the replicas (queue masks, queue bytes) and, behind the dictionary, the queues that travel as lists: round 1
  hipLaunchKernelGGL(publish_kernel, dim3(publish_grid(h)), dim3(BLOCK), 0, h->stream, h->d, t);
  rc = finish_phase(h, counts);
  if (rc) return rc;
  for (uint32_t p = 0; p < h->d.n_shards; ++p) counts[h->d.n_shards + p] = 0;     // (round 2's counts: phase 2)
  h->shard_phase = 1;
  return SWIMSIM_OK;
}

int swimsim_shard_phase2(swimsim_t* h, const uint32_t* r_counts_in, uint32_t* counts) {
  int rc = shard_check(h, 1);
  if (rc) return rc;
  if (!r_counts_in || !counts) return SWIMSIM_ERR_INVALID;
  HIPCHK(h, hipSetDevice(h->device));
  if (h->d.C) {
    // a bounded shard: one period of failureDetector for its members against the gathered bytes; the deliveries to members of
    // other shards routed into per-owner segments (kind 1: 16-byte records {dst, src, -, -}) for round 2
    const uint32_t t = (uint32_t)h->tick, tk = tick_key(h->cfg.seed, t);
    if (h->timing && !h->tick_ev[0]) for (int k = 0; k < 3; ++k) HIPCHK(h, hipEventCreate(&h->tick_ev[k]));
    if (h->timing) (void)hipEventRecord(h->tick_ev[0], h->stream);
    launch_sparse_probe(h, t, tk);
    if (h->timing) (void)hipEventRecord(h->tick_ev[1], h->stream);
    hipLaunchKernelGGL(sp_route_kernel, dim3(std::min<uint32_t>(1024u, (h->d.N + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, h->stream, h->d);
    rc = finish_phase(h, counts);
    if (rc) return rc;
    for (uint32_t p = 0; p < h->d.n_shards; ++p) { counts[p] = 0; counts[2 * h->d.n_shards + p] = 0; }
    if (h->timing) { float a = 0; HIPCHK(h, hipEventElapsedTime(&a, h->tick_ev[0], h->tick_ev[1])); h->probe_ms += a; }
    h->shard_phase = 2;
    return SWIMSIM_OK;
  }
  // the peers' dictionaries into my numbering, then one period of failureDetector for my members against the replicas; the
  // deliveries for members of other shards leave the kernel routed by owner (kind 1: 8-byte records {dst, src}) for round 2
  const uint32_t t = (uint32_t)h->tick, tk = tick_key(h->cfg.seed, t);
  if (h->timing && !h->tick_ev[0]) for (int k = 0; k < 3; ++k) HIPCHK(h, hipEventCreate(&h->tick_ev[k]));
  hipLaunchKernelGGL(xlat_kernel, dim3(h->d.n_shards + XLAT_INDEX_BLOCKS), dim3(BLOCK), 0, h->stream, h->d, t, peer_counts(h, r_counts_in), PeerView{});
  if (h->timing) (void)hipEventRecord(h->tick_ev[0], h->stream);
  launch_probe(h, t, tk, 0u);
  if (h->timing) (void)hipEventRecord(h->tick_ev[1], h->stream);
  rc = finish_phase(h, counts);
  if (rc) return rc;
  for (uint32_t p = 0; p < h->d.n_shards; ++p) counts[p] = 0;                       // (round 1 is over)
  if (h->timing) { float a = 0; HIPCHK(h, hipEventElapsedTime(&a, h->tick_ev[0], h->tick_ev[1])); h->probe_ms += a; }
  h->shard_phase = 2;
  return SWIMSIM_OK;
}

int swimsim_shard_phase3(swimsim_t* h, const uint32_t* p_counts_in, const uint32_t* x_counts_in) {
  int rc = shard_check(h, 2);
  if (rc) return rc;
  if (!p_counts_in || !x_counts_in) return SWIMSIM_ERR_INVALID;
  HIPCHK(h, hipSetDevice(h->device));
  if (h->d.C) {
    const uint32_t t = (uint32_t)h->tick, tk = tick_key(h->cfg.seed, t);
    hipLaunchKernelGGL(sp_ingest_kernel, dim3(std::min<uint32_t>(1024u, (h->d.N + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, h->stream, h->d, t, peer_counts(h, p_counts_in), 0u);
    if (h->timing) (void)hipEventRecord(h->tick_ev[1], h->stream);
    launch_sparse_merge(h, t, tk);
    if (h->timing) (void)hipEventRecord(h->tick_ev[2], h->stream);
    rc = finish_phase(h, nullptr);
    if (rc) return rc;
    if (h->timing) { float b = 0; HIPCHK(h, hipEventElapsedTime(&b, h->tick_ev[1], h->tick_ev[2])); h->merge_ms += b; h->timed_ticks++; }
    h->tick++;
    h->shard_phase = 0;
    return SWIMSIM_OK;
  }
  const uint32_t t = (uint32_t)h->tick;
  hipLaunchKernelGGL(ingest_kernel, dim3(ingest_grid(h)), dim3(BLOCK), 0, h->stream, h->d, t, peer_counts(h, p_counts_in), PeerView{});
  if (h->timing) (void)hipEventRecord(h->tick_ev[1], h->stream);
  const bool rk = records_kernel_every_tick(h);
  if (rk) hipLaunchKernelGGL(records_kernel, dim3(h->d.nblocks), dim3(BLOCK), 0, h->stream, h->d, t);
  hipLaunchKernelGGL(merge_kernel, dim3(h->d.nblocks), dim3(BLOCK), 0, h->stream, SWIM_STATE_ARG(h), t, rk ? 0u : 1u, 0u);
  if (h->timing) (void)hipEventRecord(h->tick_ev[2], h->stream);
  if (h->d.G) hipLaunchKernelGGL(settle_publish_kernel, dim3(1), dim3(BLOCK), 0, h->stream, h->d, t);
  rc = finish_phase(h, nullptr);
  if (rc) return rc;
  if (h->timing) { float b = 0; HIPCHK(h, hipEventElapsedTime(&b, h->tick_ev[1], h->tick_ev[2])); h->merge_ms += b; h->timed_ticks++; }
  h->tick++;
  h->shard_phase = h->d.G ? 3 : 0;                // settling: the tick ends with round 3 + swimsim_shard_settle_commit
  return SWIMSIM_OK;
}

int swimsim_shard_gather_buffers(swimsim_t* h, void** send /*[2]*/, void** recv /*[2]*/, uint32_t* n_local) {
  if (!h || !send || !recv) return SWIMSIM_ERR_INVALID;
  if (h->d.C && h->d.n_shards > 1) {                // bounded handles: the queue lines (64-byte records) and the member bytes
    if (n_local) *n_local = h->d.N;
    send[0] = h->d.sp_qall + (size_t)h->d.lo * PB_SLOTS; send[1] = h->d.mb + h->d.lo;
    recv[0] = h->d.sp_qall; recv[1] = h->d.mb;
    return SWIMSIM_OK;
  }
  const bool sh = h->d.n_shards > 1;               // dense shards: the queue masks (8 bytes) and the queue bytes
  if (n_local) *n_local = sh ? h->d.N : 0u;
  send[0] = sh ? (void*)(h->d.mask_all + h->d.lo) : nullptr; send[1] = sh ? (void*)(h->d.q_all + h->d.lo) : nullptr;
  recv[0] = h->d.mask_all; recv[1] = h->d.q_all;
  return SWIMSIM_OK;
}

int swimsim_shard_settle_buffers(swimsim_t* h, void** send, void** recv, uint32_t* cap) {
  if (!h || !send || !recv) return SWIMSIM_ERR_INVALID;
  *send = h->d.s_send; *recv = h->d.s_recv;
  if (cap) *cap = h->d.s_cap;
  return SWIMSIM_OK;
}

int swimsim_shard_settle_counts(swimsim_t* h, uint32_t* counts) {
  int rc = shard_check(h, 3);
  if (rc) return rc;
  if (!counts) return SWIMSIM_ERR_INVALID;
  const uint32_t n = std::min(h->h_sync[G_SETTLE_SEND], h->d.s_cap);   // phase 3's copy of the globals
  for (uint32_t p = 0; p < h->d.n_shards; ++p) counts[p] = p == h->d.shard ? 0u : n;
  return SWIMSIM_OK;
}

int swimsim_shard_settle_commit(swimsim_t* h, const uint32_t* counts_in) {
  int rc = shard_check(h, 3);
  if (rc) return rc;
  if (!counts_in) return SWIMSIM_ERR_INVALID;
  HIPCHK(h, hipSetDevice(h->device));
  hipLaunchKernelGGL(settle_commit_kernel, dim3(1), dim3(BLOCK), 0, h->stream, h->d, (uint32_t)(h->tick - 1), peer_counts(h, counts_in), PeerView{});
  HIPCHK(h, hipGetLastError());
  h->settled_alive_tick = ~0ull;                  // the bases just changed: a list read_view cached between phase 3 and now is stale
  h->shard_phase = 0;
  return SWIMSIM_OK;
}

/* A whole cluster of bounded handles in ONE process (one handle per GPU, or several on one GPU): the tick loop AND the exchange
 * inside the library, enqueued on the handles' streams -- device-to-device (peer) copies ordered by events, the record counts
 * read by the receiving kernels from device memory.  No host synchronisation between the first tick and the last: what a
 * single-process host (a Haskell program with eight GPUs, tests/..., scripts/config5_cluster_one_gpu.py) calls instead of lending
 * an exchange callback to swimsim_shard_step.  hs[k] must be shard k of n bounded (view_cap) handles of one configuration with the
 * same fault schedule. */
// what must be equal across the handles of a cluster: the resolved configuration apart from the shard index and the device
// (field by field: the struct has padding -- behind num_to_gossip and n_members -- that resolve_config copies from the caller's
// struct as it finds it; a memcmp refused valid clusters whose configs were filled in on the stack)
static bool same_cluster_config(const swimsim_config_t& a, const swimsim_config_t& b) {
#define SAME(f) (a.f == b.f)
  return SAME(struct_size) && SAME(abi_version) && SAME(num_to_gossip) && SAME(gossip_interval_us) && SAME(n_members) && SAME(seed) &&
         SAME(probes_per_tick) && SAME(indirect_k) && SAME(loss_ppm) && SAME(suspicion_ticks) && SAME(retransmit_mult) &&
         SAME(max_subjects) && SAME(gc_ticks) && SAME(event_cap) && SAME(event_mask) && SAME(inbox_cap) && SAME(n_shards) &&
         SAME(target_scheme) && SAME(join_pull) && SAME(pull_ticks) && SAME(view_cap) && SAME(strict_reference_rules) && SAME(push_pull);
#undef SAME
}

/* The same for DENSE handles.  Per tick and handle, on the handle's own stream:
 *   begin_kernel + publish_kernel -> e0 | wait every peer's e0; xlat_kernel (dictionaries, list index, the peers' slices of the
 *   replicas pulled over) + probe_kernel -> e1 | wait every peer's e1; ingest_kernel reads the peers' round-2 segments and lists
 *   WHERE THEY LIE (PeerView: same device, or a peer device over xGMI) with the counts from the peers' own words -> e2;
 *   merge_kernel; with settling settle_publish -> e3 | wait every peer's e3; settle_commit -> e2 instead.
 * The next tick's begin waits for every peer's e2 (they are done with my send buffers).  Nothing is copied but the replicas,
 * nothing comes back to the host.  State pulls (join_pull, pull_ticks) split the start of the tick around round 0 as the phase calls do:
 * begin part A + pull_send -> eJ | wait every peer's eJ; the pulls from local hosts, begin part B (the peers' records read in place),
 * publish -> e0. */
static int cluster_step_dense(swimsim_t** hs, uint32_t n, uint32_t nticks) {
  for (uint32_t k = 0; k < n; ++k) {
    swimsim* h = hs[k];
    if (h->begun) return set_err(h, SWIMSIM_ERR_STATE, "cluster_step: a tick is in progress on this handle");
  }
  const bool settling = hs[0]->d.G != 0;
  std::vector<size_t> fend(n, 0), fpos(n, 0);
  std::vector<std::vector<hipEvent_t>> ev(n, std::vector<hipEvent_t>(5, nullptr));   // e0 .. e3 (above), [4] = eJ: round 0 published
  auto cleanup = [&]() { for (auto& e : ev) for (hipEvent_t x : e) if (x) (void)hipEventDestroy(x); };
  auto broken = [&](swimsim* h, hipError_t e, const char* what) -> int {
    for (uint32_t k = 0; k < n; ++k) { hs[k]->poisoned = true; (void)hipSetDevice(hs[k]->device); (void)hipStreamSynchronize(hs[k]->stream); }
    cleanup();
    return set_err(h, SWIMSIM_ERR_DEVICE, std::string("cluster_step: ") + what + ": " + hipGetErrorString(e) + " (the cluster's handles are poisoned)");
  };
#define CCHK(h, call) do { const hipError_t e_ = (call); if (e_ != hipSuccess) return broken((h), e_, #call); } while (0)
  // handles on different devices: the kernels read the peers' buffers over xGMI -- peer access both ways
  for (uint32_t k = 0; k < n; ++k)
    for (uint32_t p = 0; p < n; ++p) {
      if (hs[k]->device == hs[p]->device) continue;
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, hs[k]->device, hs[p]->device) != hipSuccess || !can)
        return set_err(hs[k], SWIMSIM_ERR_DEVICE, "cluster_step: device " + std::to_string(hs[k]->device) + " cannot access device " + std::to_string(hs[p]->device) + " (peer access)");
      (void)hipSetDevice(hs[k]->device);
      const hipError_t e = hipDeviceEnablePeerAccess(hs[p]->device, 0);
      if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return set_err(hs[k], SWIMSIM_ERR_DEVICE, std::string("cluster_step: hipDeviceEnablePeerAccess: ") + hipGetErrorString(e));
      (void)hipGetLastError();
    }
  std::vector<PeerView> pv(n);
  std::vector<JoinView> jv(n);
  for (uint32_t k = 0; k < n; ++k) {
    swimsim* h = hs[k];
    { const hipError_t e_ = hipSetDevice(h->device); if (e_ != hipSuccess) { cleanup(); return set_err(h, SWIMSIM_ERR_DEVICE, hipGetErrorString(e_)); } }
    { int rc_ = upload_faults(h, nticks, &fend[k]); if (rc_) { cleanup(); return rc_; } }
    for (int e = 0; e < 5; ++e) { const hipError_t e_ = hipEventCreateWithFlags(&ev[k][e], hipEventDisableTiming); if (e_ != hipSuccess) { cleanup(); return set_err(h, SWIMSIM_ERR_DEVICE, hipGetErrorString(e_)); } }
    if (h->timing) while (h->ev_pool.size() < (size_t)nticks * 3) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) { cleanup(); return set_err(h, SWIMSIM_ERR_DEVICE, "hipEventCreate"); } h->ev_pool.push_back(e); }
    PeerView& v = pv[k];
    v = PeerView{};
    v.direct = 1u;
    for (uint32_t p = 0; p < n; ++p) {
      const DevState& q = hs[p]->d;
      v.r[p] = q.r_send; v.rn[p] = q.g + G_XLINES;
      v.q[p] = q.q_send + (size_t)k * q.p_cap; v.qn[p] = q.send_cnt + MAX_SHARDS + k;
      v.mask[p] = q.mask_all; v.qb[p] = q.q_all;
      v.st[p] = q.s_send ? q.s_send + (size_t)k * q.s_cap : nullptr; v.stn[p] = q.g + G_SETTLE_SEND;
      jv[k].direct = 1u;
      jv[k].jl[p] = q.j_send ? q.j_send + (size_t)k * q.j_cap : nullptr; jv[k].jn[p] = q.g + G_JSEND + k;
    }
  }
  for (uint32_t tck = 0; tck < nticks; ++tck) {
    const uint32_t t = (uint32_t)hs[0]->tick, tk = tick_key(hs[0]->cfg.seed, t);
    // state pulls (join_pull: a tick in which members come up; pull_ticks: every tick) split the start of the tick around
    // exchange round 0, as swimsim_shard_phase0 / phase1 do: faults, then the owners of the pull hosts write what the hosts hold for
    // the pullers' owners, then -- behind everybody's round 0 -- the pulls from local hosts, the peers' records (read in place) and
    // the rest of the start of the tick.  The schedule is replicated: every handle takes the same branch.
    std::vector<size_t> f0s(n);
    std::vector<bool> injs(n, false);
    bool round0 = hs[0]->d.pull_T != 0 && (uint32_t)(t % hs[0]->d.pull_T) < hs[0]->d.NT;
    for (uint32_t k = 0; k < n; ++k) {
      swimsim* h = hs[k];
      f0s[k] = fpos[k];
      while (fpos[k] < fend[k] && h->faults[fpos[k]].tick <= t) ++fpos[k];
      if (h->d.join_pull) for (size_t f = f0s[k]; f < fpos[k]; ++f) round0 = round0 || h->faults[f].up != 0;
    }
    for (uint32_t k = 0; k < n; ++k) {                 // the start of the tick (or its first part), my slice of the replicas
      swimsim* h = hs[k];
      CCHK(h, hipSetDevice(h->device));
      const size_t f0 = f0s[k];
      const uint32_t nf = (uint32_t)(fpos[k] - f0);
      if (tck) for (uint32_t p = 0; p < n; ++p) if (p != k) CCHK(h, hipStreamWaitEvent(h->stream, ev[p][2], 0));   // the peers are done with my send buffers
      bool inj = false;                                // messages from outside (swimsim_inject_rumor) go into the first tick's inboxes
      if (!h->injections.empty()) { const int rc_ = flush_injections(h, t, &inj); if (rc_) { for (uint32_t q = 0; q < n; ++q) hs[q]->poisoned = hs[q]->poisoned || tck != 0 || k != 0; cleanup(); return rc_; } }   // (k != 0: handles 0..k-1 are past begin_kernel of this tick)
      injs[k] = inj;
      if (round0) {
        hipLaunchKernelGGL(begin_kernel, dim3(1), dim3(BLOCK), 0, h->stream, h->d, t, tk, h->d_faults + f0, nf, h->d_joined, 1u, PeerCounts{}, JoinView{});
        const uint32_t T = h->d.pull_T, first = T ? t % T : 0u;
        const uint32_t items = nf + ((T && first < h->d.NT) ? (h->d.NT - first + T - 1u) / T : 0u);
        hipLaunchKernelGGL(pull_send_kernel, dim3(std::max(1u, std::min<uint32_t>(1024u, (items + BLOCK - 1) / BLOCK))), dim3(BLOCK), 0, h->stream,
                           h->d, t, tk, h->d_faults + f0, nf, h->d_joined);
        CCHK(h, hipEventRecord(ev[k][4], h->stream));
        continue;
      }
      hipLaunchKernelGGL(begin_kernel, dim3(1), dim3(BLOCK), 0, h->stream, h->d, t, tk, h->d_faults + f0, nf, h->d_joined, inj ? 7u : 3u, PeerCounts{}, JoinView{});
      hipLaunchKernelGGL(publish_kernel, dim3(publish_grid(h)), dim3(BLOCK), 0, h->stream, h->d, t);
      CCHK(h, hipEventRecord(ev[k][0], h->stream));
    }
    if (round0)
      for (uint32_t k = 0; k < n; ++k) {               // round 0 (read in place) + the rest of the start of the tick
        swimsim* h = hs[k];
        CCHK(h, hipSetDevice(h->device));
        const size_t f0 = f0s[k];
        const uint32_t nf = (uint32_t)(fpos[k] - f0);
        for (uint32_t p = 0; p < n; ++p) if (p != k) CCHK(h, hipStreamWaitEvent(h->stream, ev[p][4], 0));
        uint32_t nup = 0;
        if (h->d.join_pull) for (size_t f = f0; f < fpos[k]; ++f) nup += h->faults[f].up != 0;
        const uint32_t T = h->d.pull_T, first = T ? (t % T + T - h->d.lo % T) % T : 0u;       // my first periodic puller, as a local index
        const uint32_t npp = (T && first < h->d.N) ? (h->d.N - first + T - 1u) / T : 0u;
        if (nup + npp) hipLaunchKernelGGL(join_pull_kernel, dim3(std::min(nup + npp, 4096u)), dim3(BLOCK), 0, h->stream, h->d, t, tk, h->d_faults + f0, nf, h->d_joined, nup);
        if (h->d.push_pull && npp) hipLaunchKernelGGL(push_kernel, dim3(std::min(npp, 4096u)), dim3(BLOCK), 0, h->stream, h->d, t, tk, h->d_faults + f0, nf);
        hipLaunchKernelGGL(begin_kernel, dim3(1), dim3(BLOCK), 0, h->stream, h->d, t, tk, h->d_faults + f0, nf, h->d_joined, (2u | 8u) | (injs[k] ? 4u : 0u), PeerCounts{}, jv[k]);
        hipLaunchKernelGGL(publish_kernel, dim3(publish_grid(h)), dim3(BLOCK), 0, h->stream, h->d, t);
        CCHK(h, hipEventRecord(ev[k][0], h->stream));
      }
    for (uint32_t k = 0; k < n; ++k) {                 // round 1 (read in place) + the probes
      swimsim* h = hs[k];
      CCHK(h, hipSetDevice(h->device));
      for (uint32_t p = 0; p < n; ++p) if (p != k) CCHK(h, hipStreamWaitEvent(h->stream, ev[p][0], 0));
      const uint32_t gather_blocks = std::max(1u, std::min<uint32_t>(512u, (h->d.NT - h->d.N + 4u * BLOCK - 1) / (4u * BLOCK)));
      hipLaunchKernelGGL(xlat_kernel, dim3(n + XLAT_INDEX_BLOCKS + gather_blocks), dim3(BLOCK), 0, h->stream, h->d, t, PeerCounts{}, pv[k]);
      if (h->timing) CCHK(h, hipEventRecord(h->ev_pool[(size_t)tck * 3], h->stream));
      launch_probe(h, t, tk, 0u);
      if (h->timing) CCHK(h, hipEventRecord(h->ev_pool[(size_t)tck * 3 + 1], h->stream));
      CCHK(h, hipEventRecord(ev[k][1], h->stream));
    }
    for (uint32_t k = 0; k < n; ++k) {                 // round 2 (read in place) + the end of the tick
      swimsim* h = hs[k];
      CCHK(h, hipSetDevice(h->device));
      for (uint32_t p = 0; p < n; ++p) if (p != k) CCHK(h, hipStreamWaitEvent(h->stream, ev[p][1], 0));
      hipLaunchKernelGGL(ingest_kernel, dim3(ingest_grid(h)), dim3(BLOCK), 0, h->stream, h->d, t, PeerCounts{}, pv[k]);
      if (!settling) CCHK(h, hipEventRecord(ev[k][2], h->stream));
      const bool rk = records_kernel_every_tick(h);
      if (rk) hipLaunchKernelGGL(records_kernel, dim3(h->d.nblocks), dim3(BLOCK), 0, h->stream, h->d, t);
      hipLaunchKernelGGL(merge_kernel, dim3(h->d.nblocks), dim3(BLOCK), 0, h->stream, SWIM_STATE_ARG(h), t, rk ? 0u : 1u, 0u);
      if (h->timing) CCHK(h, hipEventRecord(h->ev_pool[(size_t)tck * 3 + 2], h->stream));
      if (settling) {
        hipLaunchKernelGGL(settle_publish_kernel, dim3(1), dim3(BLOCK), 0, h->stream, h->d, t);
        CCHK(h, hipEventRecord(ev[k][3], h->stream));
      }
    }
    if (settling)
      for (uint32_t k = 0; k < n; ++k) {               // round 3: every shard's word about its rows, read in place
        swimsim* h = hs[k];
        CCHK(h, hipSetDevice(h->device));
        for (uint32_t p = 0; p < n; ++p) if (p != k) CCHK(h, hipStreamWaitEvent(h->stream, ev[p][3], 0));
        hipLaunchKernelGGL(settle_commit_kernel, dim3(1), dim3(BLOCK), 0, h->stream, h->d, t, PeerCounts{}, pv[k]);
        CCHK(h, hipEventRecord(ev[k][2], h->stream));
        h->settled_alive_tick = ~0ull;
      }
    for (uint32_t k = 0; k < n; ++k) hs[k]->tick++;
  }
  int rc = SWIMSIM_OK;
  for (uint32_t k = 0; k < n; ++k) {
    swimsim* h = hs[k];
    (void)hipSetDevice(h->device);
    h->faults.erase(h->faults.begin(), h->faults.begin() + (long)fpos[k]);
    hipError_t e1 = hipGetLastError(), e2 = hipStreamSynchronize(h->stream);
    if (e1 != hipSuccess || e2 != hipSuccess) { rc = set_err(h, SWIMSIM_ERR_DEVICE, std::string("cluster_step: ") + hipGetErrorString(e1 != hipSuccess ? e1 : e2)); continue; }
    if (h->timing) {
      for (uint32_t q = 0; q < nticks; ++q) {
        float a = 0, b = 0;
        if (hipEventElapsedTime(&a, h->ev_pool[(size_t)q * 3], h->ev_pool[(size_t)q * 3 + 1]) == hipSuccess &&
            hipEventElapsedTime(&b, h->ev_pool[(size_t)q * 3 + 1], h->ev_pool[(size_t)q * 3 + 2]) == hipSuccess) { h->probe_ms += a; h->merge_ms += b; }
      }
      h->timed_ticks += nticks;
    }
    const int rc_ = check_device_errors(h);
    if (rc_ && !rc) rc = rc_;
  }
  cleanup();
  return rc;
#undef CCHK
}

int swimsim_cluster_step(swimsim_t** hs, uint32_t n, uint32_t nticks) {
  if (!hs || n < 2 || n > (uint32_t)MAX_SHARDS) return SWIMSIM_ERR_INVALID;
  for (uint32_t k = 0; k < n; ++k) {
    swimsim* h = hs[k];
    if (!h) return SWIMSIM_ERR_INVALID;
    if (h->poisoned) return set_err(h, SWIMSIM_ERR_STATE, "handle is poisoned by an earlier capacity error");
    for (uint32_t j = 0; j < k; ++j) if (hs[j] == h) return set_err(h, SWIMSIM_ERR_INVALID, "cluster_step: the same handle twice");
    if (h->d.n_shards != n || h->d.shard != k || h->shard_phase != 0 || h->tick != hs[0]->tick || !same_cluster_config(h->cfg, hs[0]->cfg))
      return set_err(h, SWIMSIM_ERR_INVALID, "cluster_step: hs[k] must be shard k of n handles of ONE cluster configuration (seed, sizes, every option), between ticks");
  }
  if (!hs[0]->d.C) return cluster_step_dense(hs, n, nticks);
  std::vector<size_t> fend(n, 0), fpos(n, 0);
  // per handle: [0] its slice of the replicas is published, [1] its records are routed, [2] it has copied what it needs from its
  // peers' send buffers (a peer may then reuse them: the next tick's publish zeroes the counters, its route overwrites the segments)
  std::vector<std::vector<hipEvent_t>> ev(n, std::vector<hipEvent_t>(3, nullptr));
  auto cleanup = [&]() { for (auto& e : ev) for (hipEvent_t x : e) if (x) (void)hipEventDestroy(x); };
  // a HIP call that fails once the ticks are under way leaves every handle of the cluster half-stepped: all of them are poisoned
  // (SWIMSIM_ERR_STATE from then on), the streams drained, the events freed
  auto broken = [&](swimsim* h, hipError_t e, const char* what) -> int {
    for (uint32_t k = 0; k < n; ++k) { hs[k]->poisoned = true; (void)hipSetDevice(hs[k]->device); (void)hipStreamSynchronize(hs[k]->stream); }
    cleanup();
    return set_err(h, SWIMSIM_ERR_DEVICE, std::string("cluster_step: ") + what + ": " + hipGetErrorString(e) + " (the cluster's handles are poisoned)");
  };
#define CCHK(h, call) do { const hipError_t e_ = (call); if (e_ != hipSuccess) return broken((h), e_, #call); } while (0)
  for (uint32_t k = 0; k < n; ++k) {
    swimsim* h = hs[k];
    { const hipError_t e_ = hipSetDevice(h->device); if (e_ != hipSuccess) { cleanup(); return set_err(h, SWIMSIM_ERR_DEVICE, hipGetErrorString(e_)); } }
    { int rc_ = upload_faults(h, nticks, &fend[k]); if (rc_) { cleanup(); return rc_; } }
    for (int e = 0; e < 3; ++e) { const hipError_t e_ = hipEventCreate(&ev[k][e]); if (e_ != hipSuccess) { cleanup(); return set_err(h, SWIMSIM_ERR_DEVICE, hipGetErrorString(e_)); } }
  }
  const uint32_t N = hs[0]->d.N;
  auto peer_copy = [&](swimsim* dst, void* to, swimsim* src, const void* from, size_t bytes) -> hipError_t {
    return dst->device == src->device ? hipMemcpyAsync(to, from, bytes, hipMemcpyDeviceToDevice, dst->stream)
                                      : hipMemcpyPeerAsync(to, dst->device, from, src->device, bytes, dst->stream);
  };
  for (uint32_t tck = 0; tck < nticks; ++tck) {
    const uint32_t t = (uint32_t)hs[0]->tick, tk = tick_key(hs[0]->cfg.seed, t);
    // phase 1 on every shard: the tick's scheduled changes, its slice of the lines into its replica
    for (uint32_t k = 0; k < n; ++k) {
      swimsim* h = hs[k];
      CCHK(h, hipSetDevice(h->device));
      const size_t f0 = fpos[k];
      while (fpos[k] < fend[k] && h->faults[fpos[k]].tick <= t) ++fpos[k];
      if (tck) for (uint32_t p = 0; p < n; ++p) if (p != k) CCHK(h, hipStreamWaitEvent(h->stream, ev[p][2], 0));   // the peers are done with my send buffers
      if (fpos[k] > f0) hipLaunchKernelGGL(sp_begin_kernel, dim3(1), dim3(BLOCK), 0, h->stream, h->d, t, h->d_faults + f0, (uint32_t)(fpos[k] - f0));
      hipLaunchKernelGGL(sp_publish_kernel, dim3((N * PB_SLOTS + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, h->stream, h->d, t);
      CCHK(h, hipEventRecord(ev[k][0], h->stream));
    }
    // round 1, the all-gather: every shard copies every peer's slice of the queue lines and member bytes, then probes
    for (uint32_t k = 0; k < n; ++k) {
      swimsim* h = hs[k];
      CCHK(h, hipSetDevice(h->device));
      for (uint32_t p = 0; p < n; ++p) {
        if (p == k) continue;
        swimsim* q = hs[p];
        CCHK(h, hipStreamWaitEvent(h->stream, ev[p][0], 0));
        CCHK(h, peer_copy(h, h->d.sp_qall + (size_t)q->d.lo * PB_SLOTS, q, q->d.sp_qall + (size_t)q->d.lo * PB_SLOTS, (size_t)N * PB_SLOTS * sizeof(uint2)));
        CCHK(h, peer_copy(h, h->d.mb + q->d.lo, q, q->d.mb + q->d.lo, (size_t)N));
      }
      launch_sparse_probe(h, t, tk);
      hipLaunchKernelGGL(sp_route_kernel, dim3(std::min<uint32_t>(1024u, (N + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, h->stream, h->d);
      CCHK(h, hipEventRecord(ev[k][1], h->stream));
    }
    // round 2, the all-to-all-v: every peer's segment for me (its whole capacity: the true count travels as a word next to it)
    for (uint32_t k = 0; k < n; ++k) {
      swimsim* h = hs[k];
      CCHK(h, hipSetDevice(h->device));
      for (uint32_t p = 0; p < n; ++p) {
        if (p == k) continue;
        swimsim* q = hs[p];
        CCHK(h, hipStreamWaitEvent(h->stream, ev[p][1], 0));
        CCHK(h, peer_copy(h, h->d.sp_pin + p, q, q->d.send_cnt + 1 * MAX_SHARDS + k, sizeof(uint32_t)));
        CCHK(h, peer_copy(h, h->d.p_recv + (size_t)p * h->d.p_cap, q, q->d.p_send + (size_t)k * q->d.p_cap, (size_t)q->d.p_cap * sizeof(uint4)));
      }
      CCHK(h, hipEventRecord(ev[k][2], h->stream));
      hipLaunchKernelGGL(sp_ingest_kernel, dim3(std::min<uint32_t>(1024u, (N + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, h->stream, h->d, t, PeerCounts{}, 1u);
      launch_sparse_merge(h, t, tk);
    }
    for (uint32_t k = 0; k < n; ++k) hs[k]->tick++;
  }
  int rc = SWIMSIM_OK;
  for (uint32_t k = 0; k < n; ++k) {
    swimsim* h = hs[k];
    (void)hipSetDevice(h->device);
    h->faults.erase(h->faults.begin(), h->faults.begin() + (long)fpos[k]);
    hipError_t e1 = hipGetLastError(), e2 = hipStreamSynchronize(h->stream);
    if (e1 != hipSuccess || e2 != hipSuccess) { rc = set_err(h, SWIMSIM_ERR_DEVICE, std::string("cluster_step: ") + hipGetErrorString(e1 != hipSuccess ? e1 : e2)); continue; }
    const int rc_ = check_device_errors(h);
    if (rc_ && !rc) rc = rc_;
  }
  cleanup();
  return rc;
}
#undef CCHK

int swimsim_shard_step(swimsim_t* h, uint32_t nticks, swimsim_exchange_fn xchg, void* ctx) {
  if (!h || !xchg) return SWIMSIM_ERR_INVALID;
  const uint32_t G = h->d.n_shards;
  std::vector<uint32_t> out(3 * MAX_SHARDS), in(3 * MAX_SHARDS);
  for (uint32_t k = 0; k < nticks; ++k) {
    int rc, need = 0;
    if (h->d.join_pull || h->d.pull_T) {
      rc = swimsim_shard_phase0(h, out.data(), &need);
      if (rc) return rc;
      if (need) {
        std::fill(in.begin(), in.end(), 0u);
        if (xchg(ctx, 0, out.data(), in.data())) { h->poisoned = true; return set_err(h, SWIMSIM_ERR_STATE, "shard_step: the exchange callback failed in round 0 (the tick is half done: the handle is poisoned)"); }
        rc = swimsim_shard_join_ingest(h, in.data());
        if (rc) return rc;
      }
    }
    rc = swimsim_shard_phase1(h, out.data());
    if (rc) return rc;
    {                                               // round 1 also all-gathers the queue masks / lines (kind 5) and queue bytes
      for (uint32_t p = 0; p < G; ++p) {            // (kind 6): N records to every peer, counted at [G + p] and [2G + p]
        out[G + p] = p == h->d.shard ? 0u : h->d.N;
        out[2 * G + p] = p == h->d.shard ? 0u : h->d.N;
      }
    }
    std::fill(in.begin(), in.end(), 0u);
    if (xchg(ctx, 1, out.data(), in.data())) { h->poisoned = true; return set_err(h, SWIMSIM_ERR_STATE, "shard_step: the exchange callback failed in round 1 (the tick is half done: the handle is poisoned)"); }
    rc = swimsim_shard_phase2(h, in.data(), out.data());
    if (rc) return rc;
    std::fill(in.begin(), in.end(), 0u);
    if (xchg(ctx, 2, out.data(), in.data())) { h->poisoned = true; return set_err(h, SWIMSIM_ERR_STATE, "shard_step: the exchange callback failed in round 2 (the tick is half done: the handle is poisoned)"); }
    rc = swimsim_shard_phase3(h, in.data() + G, in.data() + 2 * G);
    if (rc) return rc;
    if (h->d.G) {
      std::fill(out.begin(), out.end(), 0u);
      rc = swimsim_shard_settle_counts(h, out.data());
      if (rc) return rc;
      std::fill(in.begin(), in.end(), 0u);
      if (xchg(ctx, 3, out.data(), in.data())) { h->poisoned = true; return set_err(h, SWIMSIM_ERR_STATE, "shard_step: the exchange callback failed in round 3 (the tick is half done: the handle is poisoned)"); }
      rc = swimsim_shard_settle_commit(h, in.data());
      if (rc) return rc;
    }
  }
  return SWIMSIM_OK;
}

int swimsim_shard_traffic(swimsim_t* h, uint64_t out[4]) {
  if (!h || !out || h->d.n_shards < 2) return SWIMSIM_ERR_INVALID;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  uint32_t g[G_WORDS];
  HIPCHK(h, hipMemcpy(g, h->d.g, sizeof g, hipMemcpyDeviceToHost));
  const uint32_t G = h->d.n_shards;
  uint64_t recs = 0;
  for (uint32_t p = 0; p < G; ++p) if (p != h->d.shard) recs += std::min(g[G_SEND + MAX_SHARDS + p], h->d.p_cap);
  if (h->d.C) {
    out[0] = (uint64_t)h->d.N * (PB_SLOTS * sizeof(uint2) + 1u);
    out[1] = recs * sizeof(uint4); out[2] = 0; out[3] = 0;
  } else {
    out[0] = (uint64_t)h->d.N * 9u + DICT_RECS * sizeof(uint4) + (uint64_t)g[G_XLINES] * XLINE_RECS * sizeof(uint4);
    out[1] = recs * sizeof(uint2); out[2] = std::min(g[G_SEND + MAX_SHARDS + h->d.shard], h->d.p_cap); out[3] = g[G_XLINES];
  }
  return SWIMSIM_OK;
}

/* First-detection ticks are recorded by the prober's shard: set the combined (element-wise minimum over
 * all shards) array back before digest / first_detect are read on a sharded cluster. */
int swimsim_shard_set_first_suspect(swimsim_t* h, const uint32_t* combined, size_t n) {
  if (!h || !combined || n != h->d.NT) return SWIMSIM_ERR_INVALID;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipMemcpy(h->d.first_suspect, combined, n * sizeof(uint32_t), hipMemcpyHostToDevice));
  return SWIMSIM_OK;
}

int swimsim_shard_get_first_suspect(swimsim_t* h, uint32_t* out, size_t n) {
  if (!h || !out || n != h->d.NT) return SWIMSIM_ERR_INVALID;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipMemcpy(out, h->d.first_suspect, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
  return SWIMSIM_OK;
}

int swimsim_kernel_timing_enable(swimsim_t* h, int enable) {
  if (!h) return SWIMSIM_ERR_INVALID;
  h->timing = enable != 0;
  h->probe_ms = h->merge_ms = 0; h->timed_ticks = 0;
  return SWIMSIM_OK;
}

int swimsim_kernel_timing(swimsim_t* h, double* out, size_t n) {
  if (!h || !out || n < 3) return SWIMSIM_ERR_INVALID;
  out[0] = h->probe_ms; out[1] = h->merge_ms; out[2] = (double)h->timed_ticks;
  return SWIMSIM_OK;
}

int swimsim_set_view(swimsim_t* h, uint32_t observer, uint32_t subject, uint8_t state, uint32_t incarnation) {
  if (!h || observer - h->d.lo >= h->d.N || subject >= h->d.NT || state > 2 || incarnation > INC_MAX || observer == subject)
    return SWIMSIM_ERR_INVALID;
  if (h->d.C) return set_err(h, SWIMSIM_ERR_INVALID, "set_view: not available with bounded member maps (view_cap)");
  HIPCHK(h, hipSetDevice(h->device));
  hipLaunchKernelGGL(set_view_kernel, dim3(1), dim3(64), 0, h->stream, h->d, (uint32_t)h->tick, observer, subject,
                     (incarnation << 2) | state);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return check_device_errors(h);
}

}  // extern "C"
