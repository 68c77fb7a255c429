// swim_kernels.h -- the per-tick HIP kernels (gfx950).  Integer / indexing work; no MFMA on
// this path.  The binding resource is the number of scattered L2<->fabric requests per member-tick
// (measured ~45 G requests/s whatever the ALU load: profiles/), so the layout keeps ONE gathered
// word per probe target (minfo), a delivered rumour costs the receiver one bit test on its
// known-ring (swim_device.h) instead of a view lookup, the big owner-private tables are
// position-major so that a wave's accesses share sectors, and only cross-member deliveries go
// through atomics.
// Three launches per tick:
//   begin_kernel : one block: settling bookkeeping, scheduled faults, rumour-id window head (+ id dictionary).
//   probe_kernel : one period of failureDetector / probeNode' per member (src/Core.hs:233-269),
//                  closed form of the message exchange; delivers the piggyback payloads as masks.
//   merge_kernel : owner-computes end of tick: delivered rumours, timers, state rule, piggyback queue
//                  (src/Core.hs:89-117, 127-138, 142-218).
// Sharded clusters add publish_kernel / xlat_kernel / ingest_kernel around the two exchange rounds
// (DESIGN.md section 6).
#pragma once
#include "swim_device.h"

namespace swim {

// ---- block-level counter accumulation (one row per block, no global atomics) -------------
struct BlockCounters {
  unsigned v[C_COUNT];
  unsigned long long evd;
};

__device__ inline void ctr_init(BlockCounters* sh) {
  if (threadIdx.x < C_COUNT) sh->v[threadIdx.x] = 0;
  if (threadIdx.x == 0) sh->evd = 0;
  __syncthreads();
}

__device__ inline void ctr_add(BlockCounters* sh, int which, unsigned x) {
  if (x) atomicAdd(&sh->v[which], x);
}

// Block barrier that orders LDS traffic only.  __syncthreads() also waits until every global store and atomic the
// wave has in flight is acknowledged (s_waitcnt vmcnt(0)): after a kernel's scattered stores that is thousands
// of clocks per wave (profiles/r02u_sections.txt) for data nobody in the block reads.
__device__ inline void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
// the same between the lanes of one wave (LDS executes a wave's instructions in order)
__device__ inline void lds_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
}

// Sum over the 64 lanes of a wave, returned to every lane.  ALL 64 lanes must be executing (call at wave-uniform
// points).  Hand-written DPP prefix steps + one lane read: left to the compiler, a per-lane atomicAdd on one LDS
// word becomes a scalar loop over the active lanes (~6 SALU instructions x 64 per counter -- most of the 2 300
// scalar instructions a wave of merge_kernel used to execute, profiles/r02a_pmc_summary.txt).
#define SWIM_DPP(v, ctrl, rows) ((unsigned)__builtin_amdgcn_update_dpp(0, (int)(v), ctrl, rows, 0xf, false))
__device__ inline unsigned wave_sum(unsigned x) {
  x += SWIM_DPP(x, 0x111, 0xf);        // row_shr:1
  x += SWIM_DPP(x, 0x112, 0xf);        // row_shr:2
  x += SWIM_DPP(x, 0x114, 0xf);        // row_shr:4
  x += SWIM_DPP(x, 0x118, 0xf);        // row_shr:8   lane 15 of each row of 16 = the row's sum
  x += SWIM_DPP(x, 0x142, 0xa);        // row_bcast:15 into rows 1 and 3
  x += SWIM_DPP(x, 0x143, 0xc);        // row_bcast:31 into rows 2 and 3: lane 63 = the wave's sum
  return (unsigned)__builtin_amdgcn_readlane((int)x, 63);
}
__device__ inline unsigned long long wave_sum64(unsigned long long x) {
#define SWIM_DPP64(ctrl, rows) x += (unsigned long long)SWIM_DPP((uint32_t)x, ctrl, rows) | ((unsigned long long)SWIM_DPP((uint32_t)(x >> 32), ctrl, rows) << 32)
  SWIM_DPP64(0x111, 0xf); SWIM_DPP64(0x112, 0xf); SWIM_DPP64(0x114, 0xf); SWIM_DPP64(0x118, 0xf);
  SWIM_DPP64(0x142, 0xa); SWIM_DPP64(0x143, 0xc);
#undef SWIM_DPP64
  return (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)x, 63) |
         ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(x >> 32), 63) << 32);
}
// OR over the 64 lanes of a wave, in every lane (all lanes executing): the DPP steps of wave_sum with | (lanes without a source
// read 0, the identity of both)
__device__ inline unsigned long long wave_or64(unsigned long long x) {
#define SWIM_DPP64(ctrl, rows) x |= (unsigned long long)SWIM_DPP((uint32_t)x, ctrl, rows) | ((unsigned long long)SWIM_DPP((uint32_t)(x >> 32), ctrl, rows) << 32)
  SWIM_DPP64(0x111, 0xf); SWIM_DPP64(0x112, 0xf); SWIM_DPP64(0x114, 0xf); SWIM_DPP64(0x118, 0xf);
  SWIM_DPP64(0x142, 0xa); SWIM_DPP64(0x143, 0xc);
#undef SWIM_DPP64
  return (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)x, 63) |
         ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(x >> 32), 63) << 32);
}
// inclusive prefix sum over the lanes of a wave (the DPP steps of wave_sum leave it in every lane)
__device__ inline unsigned wave_prefix_incl(unsigned x) {
  x += SWIM_DPP(x, 0x111, 0xf); x += SWIM_DPP(x, 0x112, 0xf); x += SWIM_DPP(x, 0x114, 0xf); x += SWIM_DPP(x, 0x118, 0xf);
  x += SWIM_DPP(x, 0x142, 0xa); x += SWIM_DPP(x, 0x143, 0xc);
  return x;
}
// a per-lane count into the block's counter: one LDS atomic per wave (wave-uniform call)
__device__ inline void ctr_add_wave(BlockCounters* sh, int which, unsigned x) {
  const unsigned tot = wave_sum(x);
  if ((threadIdx.x & 63u) == 0u && tot) atomicAdd(&sh->v[which], tot);
}
__device__ inline unsigned wave_max(unsigned x) {            // lane 0 holds the result
  for (int o = 32; o > 0; o >>= 1) { const unsigned y = __shfl_down(x, o, 64); x = y > x ? y : x; }
  return x;
}

// the maximum over the 64 lanes of a wave, in every lane (all lanes executing)
__device__ inline unsigned wave_max_all(unsigned x) {
  for (int o = 32; o > 0; o >>= 1) { const unsigned y = __shfl_down(x, o, 64); x = y > x ? y : x; }
  return (unsigned)__builtin_amdgcn_readlane((int)x, 0);
}

__device__ inline void ctr_flush(const DevState& s, BlockCounters* sh, uint32_t row) {
  lds_barrier();
  if (threadIdx.x < C_COUNT) {
    unsigned long long x = threadIdx.x == C_EVDIGEST ? sh->evd : (unsigned long long)sh->v[threadIdx.x];
    // the row has one writer; an atomic without return so that the wave does not wait for the old value
    if (x) atomicAdd(reinterpret_cast<unsigned long long*>(&s.blk[(size_t)row * C_COUNT + threadIdx.x]), x);
  }
}

// ---- delivery: "dst merges src's start-of-tick piggyback buffer" ----------------------------
__device__ inline void push_commit(const DevState& s, uint32_t t, uint32_t dst, uint32_t srcw, uint32_t pos) {
  if (pos < s.inbox_cap) {
    s.inbox[(size_t)dst * s.inbox_cap + pos] = srcw;
  } else {
    uint32_t o = atomicAdd(&s.g[G_OVF0 + (t & 1u)], 1u);
    if (o < s.ovf_cap) s.ovf[(size_t)(t & 1u) * s.ovf_cap + o] = make_uint2(dst, srcw);
    else atomicOr(&s.g[G_ERR], (uint32_t)ERRF_OVF);
  }
}
__device__ inline void push(const DevState& s, uint32_t t, uint32_t dst, uint32_t srcw) {
  push_commit(s, t, dst, srcw, atomicAdd(&s.inbox_cnt[dst], 1u));
}

// make sure subject j (global id) has a rumour slot (first rumour about j); returns nothing: the slot
// is usually only needed by the NEXT kernel (get_slot waits for it)
__device__ inline void ensure_slot(const DevState& s, uint32_t j) {
  uint32_t cur = s.minfo[j];
  while ((cur & MI_SLOT) == 0u) {
    const uint32_t seen = atomicCAS(&s.minfo[j], cur, cur | MI_SLOT);   // 0xFFFF = being allocated
    if (seen == cur) {
      // a reclaimed row if there is one (pushed only by settle_finish, between the tick kernels), else a new one
      uint32_t r;
      const int f = (int)atomicAdd(&s.g[G_NFREE], 0xFFFFFFFFu) - 1;
      if (f >= 0) r = s.free_rows[f];
      else {
        atomicAdd(&s.g[G_NFREE], 1u);
        r = atomicAdd(&s.g[G_NSLOTS], 1u);
        if (r >= s.R_phys) { atomicOr(&s.g[G_ERR], (uint32_t)ERRF_ROWS); r = 0; }
      }
      if (atomicAdd(&s.g[G_NLIVE], 1u) >= s.R_max) atomicOr(&s.g[G_ERR], (uint32_t)ERRF_SUBJECTS);
      s.subject_of[r] = j;
      s.slot_base[r] = s.base_key[j];
      s.slot_last[r] = NONE32;
      s.slot_born[r] = s.g[G_NRUM];                 // any rumour id about j is allocated after this point
      s.slot_used[r] = 1;
      __threadfence();
      atomicXor(&s.minfo[j], MI_SLOT ^ (r + 1u));
      mb_set_row(s, j);
      return;
    }
    cur = seen;
  }
}

// slot of subject j, allocated if need be, for callers that need it at once (payload ingest)
__device__ inline uint32_t get_slot(const DevState& s, uint32_t j) {
  ensure_slot(s, j);
  uint32_t v = s.minfo[j] & MI_SLOT;
  for (int spin = 0; spin < 4096 && (v == 0u || v == MI_SLOT); ++spin) v = atomicOr(&s.minfo[j], 0u) & MI_SLOT;
  if (v == 0u || v == MI_SLOT) { atomicOr(&s.g[G_ERR], (uint32_t)ERRF_SUBJECTS); return 0u; }
  return v - 1u;
}

// -DSWIM_SECTION_CLOCKS (measurement build, scripts/section_clocks.py): every wave adds the shader clocks it
// spent between two marks to a table behind the counter rows; nothing of it is in the product build
#ifdef SWIM_SECTION_CLOCKS
#define SECT_BEGIN(base) unsigned long long sect_t_ = clock64(); SECT_ADD((base) + 15, 1ull)
#define SECT_ADD(k, v) do { const unsigned long long b_ = __ballot(1); \
    if ((threadIdx.x & 63u) == (uint32_t)(__ffsll(b_) - 1)) atomicAdd(reinterpret_cast<unsigned long long*>(&s.blk[((size_t)s.nblocks + 1) * C_COUNT + (blockIdx.x & 63u) * 64u + (k)]), (unsigned long long)(v)); } while (0)
#define SECT(k) do { const unsigned long long n_ = clock64(); SECT_ADD(k, n_ - sect_t_); sect_t_ = clock64(); } while (0)
#define SECT_COUNT(k) do { const unsigned long long b_ = __ballot(1); \
    if ((threadIdx.x & 63u) == (uint32_t)(__ffsll(b_) - 1)) atomicAdd(reinterpret_cast<unsigned long long*>(&s.blk[((size_t)s.nblocks + 1) * C_COUNT + (blockIdx.x & 63u) * 64u + (k)]), (unsigned long long)__popcll(b_)); } while (0)
#define SECT_PARAM , unsigned long long& sect_t_
#define SECT_ARG , sect_t_
#else
#define SECT_BEGIN(base) ((void)0)
#define SECT(k) ((void)0)
#define SECT_COUNT(k) ((void)0)
#define SECT_PARAM
#define SECT_ARG
#endif
#ifndef PSTAT                   // tests/hostemu -DSWIM_PATH_STATS counts how often a site runs per lane / per wave
#define PSTAT(...) ((void)0)
#define PSITE(x) ((void)0)
#else
#define PSITE(x) (psite = (x))
#endif
// -DSWIM_STATE_BY_POINTER (measurement knob, default off): the two tick kernels take the state through a pointer to a
// device copy instead of by value.  By value, the compiler fetches every field a kernel uses at its entry (kernel
// arguments are loaded there by construction) and parks what does not fit in scalar registers in vector lanes --
// merge_kernel: 79 scalars parked, 922 lane reads at use sites (scripts/isa_histogram.py); through a pointer a
// field is a scalar load where it is used, and the fields of cold branches are never touched on the hot path.
#ifdef SWIM_STATE_BY_POINTER
#define SWIM_STATE_PARAM const DevState* __restrict__ state_ptr
#define SWIM_STATE_BIND const DevState& s = *state_ptr;
#else
#define SWIM_STATE_PARAM DevState s
#define SWIM_STATE_BIND
#endif

// ================================================================================================
// probe kernel
// ================================================================================================
// "dst merges src's start-of-tick queue" for a LOCAL source and a LOCAL destination: the mask by
// atomicOr, filtered by what dst already knows, plus an explicit record when the mask cannot carry all
// of it (swim_device.h).
__device__ inline void deliver_local(const DevState& s, uint32_t t, bool use_mask, unsigned long long stale,
                                     uint32_t dst_li, uint32_t src_li, uint32_t msrc, unsigned long long srcmask) {
  if (use_mask) {
    const unsigned long long m = srcmask & ~(s.pk[dst_li].y & ~stale);
    if (m) atomicOr(&s.inmask[dst_li], m);
  }
  if (!use_mask || (msrc & MI_OOW)) push(s, t, dst_li, mi_src(src_li, msrc));
}

struct FaultRec { uint32_t member, up; };
// A tick whose scheduled changes are a few CRASHES (the benchmarked regime: ~1 per tick) runs without begin_kernel too (`fold`,
// round 5): the tick's crash list rides into probe_kernel as an overlay on ground truth -- whatever a prober reads about a member
// on the list, it reads "down, empty queue" (what begin_kernel would have stored before the launch) --, workgroup 0 stores the
// changes on the side for merge_kernel and everything after it.  Joins are not folded (an announcement takes a rumour id: the
// window head of the tick depends on it).
constexpr uint32_t FOLD_MAX_CRASHES = 8;
struct CrashList { uint32_t n; uint32_t member[FOLD_MAX_CRASHES]; };
#ifndef SWIM_PROBE_WAVES
#define SWIM_PROBE_WAVES 5
#endif
#ifndef SWIM_PROBE_WAVES8       // occupancy asked of the wider instantiations (numToGossip 5-8 / 9-12 / 13-16)
#define SWIM_PROBE_WAVES8 3
#endif
#ifndef SWIM_PROBE_WAVES12
#define SWIM_PROBE_WAVES12 3
#endif
#ifndef SWIM_PROBE_WAVES16
#define SWIM_PROBE_WAVES16 2
#endif
constexpr int PX_KG = 4;        // proxy indices per round of the wave's indirect-probe pass (probe_kernel pass 5)
// SH: the handle is a shard of a cluster (n_shards > 1) -- targets, proxies and sources may live on other shards; what the
// kernel needs of them comes from the replicas all-gathered at the start of the tick (swim_device.h, "cross-shard exchange"):
// ground truth (minfo / mb), queue byte (q_all), queue mask (mask_all, through the owner's ring dictionary xl).  A delivery to
// a member of another shard is an 8-byte record {dst, src} for the owner of dst.  The unsharded instantiation has none of it.
template <int PMAX, bool SH>
__global__ __launch_bounds__(BLOCK, PMAX <= 4 ? SWIM_PROBE_WAVES : PMAX <= 8 ? SWIM_PROBE_WAVES8 : PMAX <= 12 ? SWIM_PROBE_WAVES12 : SWIM_PROBE_WAVES16) void probe_kernel(SWIM_STATE_PARAM, uint32_t t, uint32_t tk, Offsets off, uint32_t fold, CrashList crashes) {
  SWIM_STATE_BIND
  __shared__ BlockCounters sh;
  __shared__ uint32_t ordn;                        // deliveries handed to a shard's ingest (sharded runs)
  // the peers' ring dictionaries as far as this kernel needs them (xlat_kernel): a peer's ring position -> mine; 0xFF: my masks
  // cannot carry that rumour this tick, 0xFE: the position is empty
  __shared__ uint8_t xpos[SH ? MAX_SHARDS * DICT_ENTRIES : 1];
  __shared__ uint32_t rt_cnt[MAX_SHARDS], rt_base[MAX_SHARDS];
  __shared__ uint32_t rt_wave[SH ? MAX_SHARDS : 1][BLOCK / 64];   // a wave's Ping records per owner, then where its run starts
  if (SH) for (uint32_t k = threadIdx.x; k < s.n_shards * DICT_ENTRIES; k += BLOCK) { const uint2 e = s.xl[k]; xpos[k] = e.x == NONE32 ? (uint8_t)0xFEu : (uint8_t)(e.y >> 24); }   // (ctr_init's barrier publishes it)
  // pass 5, per wave: the (prober, proxy) pairs of a round, the probers' context, the chains' outcomes
  __shared__ uint4 px_item[BLOCK / 64][64 * PX_KG];      // {proxy, its minfo, prober lane | proxy index << 8, -}
  __shared__ uint4 px_ctx[BLOCK / 64][64];               // {prober's minfo, target, its minfo, -}
  __shared__ unsigned long long px_mask[BLOCK / 64][64]; // the prober's queue mask
  __shared__ unsigned long long px_got[BLOCK / 64][64];  // masks the prober pulled through relayed Acks
  __shared__ uint32_t px_ack[BLOCK / 64][64];            // some proxy relayed an Ack
  if (threadIdx.x == 0) ordn = 0;
  ctr_init(&sh);
  const uint32_t li = blockIdx.x * BLOCK + threadIdx.x;
  const uint32_t i = s.lo + li;                    // global id
  // the tick's crashes as an overlay on what is read about a member (a folded tick; crashes.n = 0 otherwise): down, queue gone
  auto overlay = [&](uint32_t member, uint32_t m) -> uint32_t {
    bool hit = false;
#pragma unroll
    for (uint32_t k = 0; k < FOLD_MAX_CRASHES; ++k) hit |= k < crashes.n && crashes.member[k] == member;
    return hit ? (m & ~(MI_UP | MI_PB)) : m;
  };
  // ONE round of loads at the start (round 6): the member's own word, its queue mask, and the bytes of the first draw of each of its
  // probe targets -- the draws are pure functions of (tick, member), so they need not wait for the member's own word
  const uint32_t mk = mix32(tk ^ i);
  uint32_t pre_c[PMAX], pre_b[PMAX];
  unsigned long long mymask0 = 0;
  // (the narrow instantiation only: with numToGossip = 10 the twelve draws' registers cost more than the round trip saves --
  // 346 -> 377 us per launch, profiles/r06e_ab_probe_first_draws.txt)
  const bool pre_sel = PMAX <= 4 && li < s.N && s.scheme != 1u;
#pragma unroll
  for (int p = 0; p < PMAX; ++p) { pre_c[p] = 0; pre_b[p] = 0; }
  if (pre_sel) select_first_draws<PMAX>(s, mk, s.P, pre_c, pre_b);
  if (li < s.N) mymask0 = s.pk[li].x;
  const uint32_t mi = li < s.N ? overlay(i, s.minfo[i]) : 0u;
  const bool act = mi_up(mi);
  // masks are exact only if few rumour ids appeared since they were built (swim_device.h)
  // `fold` (swimsim_step, a PLAIN tick: no scheduled change, no message from outside, no pull, no settling, one handle): there is
  // NO begin_kernel launch -- the tick's window head is the rumour-id counter as it stands (nothing allocates ids between the ticks'
  // merge kernels), the head before it is what the last tick left in G_HEAD, and workgroup 0 leaves what merge_kernel needs of the
  // start of the tick (below): a launch and a kernel boundary less per tick (~ 11 us; profiles/r03i_*: a boundary is ~ 6 us)
  const uint32_t Hprev = fold ? s.g[G_HEAD] : s.g[G_PREV], H = fold ? s.g[G_NRUM] : s.g[G_HEAD];
  if (fold && blockIdx.x == 0) {
    // begin_kernel's part B for a plain tick, by the one workgroup: nothing of it is read by this kernel's other workgroups
    // (they took H and Hprev from words nobody writes during this launch), all of it by merge_kernel behind the kernel boundary
    if (threadIdx.x == 0) {
      s.g[G_PREV_NEW] = Hprev; s.g[G_HEAD_NEW] = H;
      s.g[G_RIDS_OFF] = (H - Hprev > RID_MASK + 1u - RID_NEAR - KW_BITS) ? 1u : 0u;
    }
    // the tick's crashes, as begin_kernel stores them (a member that is down already: nothing; first_suspect needs no reset --
    // it is reset when a member comes up and only ever set while it is down)
    // -- with atomics: the probers of a member that crashes in this very tick give it a view row in this launch (ensure_slot:
    // a CAS on the same word, an atomicOr on the byte's word)
    if (threadIdx.x < crashes.n) {
      const uint32_t mbr = crashes.member[threadIdx.x];
      const uint32_t m0 = atomicAnd(&s.minfo[mbr], ~(MI_UP | MI_PB));
      atomicAnd(reinterpret_cast<uint32_t*>(s.mb) + (mbr >> 2), ~((MB_UP | (0xFu << MB_PBN_SHIFT) | MB_OOW) << (8u * (mbr & 3u))));
      if (mi_up(m0)) {
        s.crash_tick[mbr] = t;
        s.pk[mbr - s.lo].x = 0ull;
        s.inbox_cnt[mbr - s.lo] = 0u;
      }
    }
    for (uint32_t k = threadIdx.x; k <= TODO_REGIONS; k += blockDim.x) s.todo_n[k * 16u] = 0;
    if (t)
      for (uint32_t k = threadIdx.x; k < s.tovf_nsub; k += blockDim.x)
        s.tovf_n[((((t - 1u) % s.S) * 2u + ((((t - 1u) / s.S) & 1u) ^ 1u)) * s.tovf_nsub + k) * 16u] = 0;
    if (threadIdx.x < KN_BITS) {
      const uint2 r = s.rum[rid_at(threadIdx.x, H) & RID_MASK];
      const uint32_t row = r.x < s.R_phys ? r.x : 0u;
      s.ring[threadIdx.x] = make_uint4(r.x, r.y, s.slot_base[row], s.subject_of[row]);
    }
  }
  // (strict reference rules: never -- a rumour the literal rule ignored may be accepted later, so no delivery may be filtered as
  // "known already": every delivery is an explicit record, every entry of its line is examined; include/swimsim.h)
  const bool use_mask = H - Hprev <= MASK_SLACK && !s.strict;
  const unsigned long long stale = stale_positions(Hprev, H);   // ring positions nobody may trust this tick
  unsigned n_pings = 0;
  unsigned payloads = 0, rumors = 0, dfail = 0, preqs = 0, susp = 0, fsusp = 0;
  unsigned long long ackacc = 0;                  // masks this member pulls in with its Acks
  bool wrote_rec = false;                          // this member left an explicit record somewhere: the records phase of merge_kernel has work
  SECT_BEGIN(32);
  // what passes 1-4 leave for the wave's pass 5 (indirect probes) and for the outputs
  const uint32_t mycnt = mi_pbn(mi);
  unsigned long long mymask = 0;
  uint32_t picks[PMAX], pinfo[PMAX];
  uint32_t failmask = 0;                           // probe indices that ended without an Ack (unlessAck; D2, D3)
  uint32_t pingmask = 0;                           // sharded: probe indices whose Ping carries my queue to a REMOTE target: records {target, me}
  unsigned long long pingown = 0;                  //   ... and the owners of those targets, 4 bits per probe index
  uint32_t nfail = 0, nack = 0;
#pragma unroll
  for (int p = 0; p < PMAX; ++p) { picks[p] = 0; pinfo[p] = 0; }
  // "dst merges src's start-of-tick queue", left to the ingest of dst's owner: dst on another shard,
  // or a remote src whose queue takes more than a mask translation (then the owner may be this very shard)
  auto emit_order = [&](uint32_t dst, uint32_t src) {
    const uint32_t pos = atomicAdd(&ordn, 1u);
    if (pos < s.ord_cap) s.ord[(size_t)blockIdx.x * s.ord_cap + pos] = make_uint2(dst, src);
    else atomicOr(&s.g[G_ERR], (uint32_t)ERRF_XCHG);
  };
  // the queue of REMOTE member src as a mask over MY ring: its replicated mask through its owner's dictionary.  false: the
  // queue travels as a list this tick, or holds an entry my masks cannot carry (an id younger than my head) -- then the
  // delivery goes to ingest_kernel, which has the machinery (foreign lines, explicit records)
  auto pull_remote = [&](uint32_t src, uint32_t qsrc, unsigned long long m, unsigned long long* bits) -> bool {   // m = mask_all[src]
    if (!SH) return false;
    if ((qsrc & Q_OOW) || !use_mask) return false;
    unsigned long long acc = 0;
    const uint8_t* d = xpos + owner_of(s, src) * DICT_ENTRIES;
    while (m) {
      const uint32_t q = (uint32_t)__ffsll((unsigned long long)m) - 1u;
      m &= m - 1ull;
      const uint32_t at = d[q];
      if (at == 0xFEu) continue;
      if (at == 0xFFu) return false;
      acc |= 1ull << at;
    }
    *bits |= acc;
    return true;
  };
  if (act) {
    mymask = (mycnt && use_mask) ? mymask0 : 0ull;
    bool valid[PMAX];                               // probe index p is in use this period
    const bool robust = s.scheme == 1u;
    // the robust scheme's Ping payloads are PULLED by the target (its pingers are computable) -- on one handle; on
    // a sharded cluster the pinger may live elsewhere, and the payloads are pushed like the random scheme's
    const bool pull = robust && s.n_shards == 1u;
    uint32_t np;                                    // probe indices in play
    if (!robust) {
      // ms <- kRandomMembers store (numToGossip cfg) []        (src/Core.hs:239)
      np = select_members<PMAX>(s, mk, i, s.P, P_SELECT, 0, nullptr, 0, picks, pinfo, use_mask, nullptr, pre_sel, pre_c, pre_b);
      if (crashes.n) {
#pragma unroll
        for (int p = 0; p < PMAX; ++p) pinfo[p] = overlay(picks[p], pinfo[p]);
      }
      n_pings = np;
#pragma unroll
      for (int p = 0; p < PMAX; ++p) valid[p] = (uint32_t)p < np;
    } else {
      // the "robust scheme" (FIXME at src/Core.hs:232): probe p goes to (i + o(t,p)) mod N -- a rotation
      // shared by everybody, so that a member's pingers are known to it; targets that are not Alive in
      // my view are skipped (include/swimsim.h)
      np = s.P;
#pragma unroll
      for (int p = 0; p < PMAX; ++p) {
        valid[p] = false; picks[p] = 0; pinfo[p] = 0;
        if ((uint32_t)p < np && off.o[p]) {
          uint32_t c = i + off.o[p]; if (c >= s.NT) c -= s.NT;
          picks[p] = c; pinfo[p] = overlay(c, probe_mi(s, c, use_mask));
          valid[p] = view_alive(s, li, pinfo[p]);
          n_pings += valid[p] ? 1u : 0u;
        }
      }
    }
    SECT(32);                                       // target selection
    // Passes 1-4 run over the probe indices CH at a time: what a pass keeps per probe -- outcomes, the 16-byte `pk` of the
    // target, inbox positions -- is live for CH probes only.  With every index of a 12-wide kernel live at once (the
    // reference's default numToGossip = 10, src/Util.hs:48) the kernel held 162 registers and ran at 2 waves per SIMD.
    constexpr int CH = PMAX <= 4 ? PMAX : 4;
#pragma unroll
    for (int c0 = 0; c0 < PMAX; c0 += CH) {
    // pass 1: outcome of every direct probe -- pure arithmetic on the gathered info words.
    //   Direct (Ping seq j) is delivered iff not lost and j is up (src/Core.hs:246);
    //   j answers Ack (src/Core.hs:97-99), which may be lost too.
    bool ping_ok[CH], ack_ok[CH];
#pragma unroll
    for (int q = 0; q < CH; ++q) {
      const int p = c0 + q;
      ping_ok[q] = false; ack_ok[q] = false;
      if (valid[p]) {
        ping_ok[q] = mi_up(pinfo[p]) && !lost(s, tk, P_L_PING, i, picks[p], p);
        ack_ok[q] = ping_ok[q] && !lost(s, tk, P_L_ACK, picks[p], i, p);
      }
    }
    // pass 2: one 16-byte gather per reached LOCAL target, issued together: its queue mask (the Ack's
    // payload, pulled by the prober itself) and its known-ring (what my Ping's payload can still tell it)
    ulonglong2 tk2[CH];
    uint32_t qrem[CH];                              // sharded: the replicated queue byte of a remote target whose Ack arrives
#pragma unroll
    for (int q = 0; q < CH; ++q) {
      const int p = c0 + q;
      tk2[q] = make_ulonglong2(0ull, 0ull); qrem[q] = 0u;
      if (use_mask && ping_ok[q] && (!SH || is_local(s, picks[p])) && (mymask || (ack_ok[q] && mi_pbn(pinfo[p]))) && !ABL(ABL_PK_GATHER))
        tk2[q] = s.pk[picks[p] - s.lo];
      // ... and of a remote target: byte and mask from the replicas, in the same round of loads (one after the other per probe
      // they were two dependent round trips x P: 30 us of a 90 us launch at 524 288 members, profiles/r05d_*)
      if (SH && ack_ok[q] && !is_local(s, picks[p])) { qrem[q] = s.q_all[picks[p]]; tk2[q].x = s.mask_all[picks[p]]; }
    }
    SECT(33);                                       // outcomes + the targets' pk gathers issued
    // pass 3: the Pings' piggyback payloads: at most one atomicOr per target
    if (mycnt) {
      uint32_t pos[CH];
      const bool expl = !use_mask || (mi & MI_OOW);
#pragma unroll
      for (int q = 0; q < CH; ++q) {
        const int p = c0 + q;
        pos[q] = 0;
        if (ping_ok[q]) {
          payloads++; rumors += mycnt;
          if (pull) continue;                        // the target pulls it (below): its pingers are computable
          if (!SH || is_local(s, picks[p])) {
            const uint32_t dl = picks[p] - s.lo;
            const unsigned long long m = mymask & ~(tk2[q].y & ~stale);   // only what the target does not know
            if (m && !ABL(ABL_PUSH_ATOMIC)) atomicOr(&s.inmask[dl], m);
            if (expl) { pos[q] = atomicAdd(&s.inbox_cnt[dl], 1u); wrote_rec = true; }
          } else if (SH) {
            pingmask |= 1u << p;                     // the target's owner delivers it (from its replica of my queue): a record, written at the end
            pingown |= (unsigned long long)owner_of(s, picks[p]) << (4 * p);
          }
        }
      }
      if (expl && !pull) {
#pragma unroll
        for (int q = 0; q < CH; ++q)
          if (ping_ok[q] && (!SH || is_local(s, picks[c0 + q]))) push_commit(s, t, picks[c0 + q] - s.lo, mi_src(li, mi), pos[q]);
      }
    }
    SECT(34);                                       // pushes
    if (pull) {
      // the Pings that reach ME this period: probe p of member q = i - o(t,p), if q is up, sees me Alive
      // and the Ping is not lost.  I merge q's queue: a gather instead of q's atomicOr.
#pragma unroll
      for (int qq = 0; qq < CH; ++qq) {
        const int p = c0 + qq;
        if ((uint32_t)p >= s.P || !off.o[p]) continue;
        uint32_t q = i + s.NT - off.o[p]; if (q >= s.NT) q -= s.NT;
        const uint32_t mq = overlay(q, probe_mi(s, q, use_mask));
        if (!mi_up(mq) || !mi_pbn(mq) || !view_alive(s, q - s.lo, mi) || lost(s, tk, P_L_PING, q, i, p)) continue;
        if (use_mask) ackacc |= s.pk[q - s.lo].x;
        if (!use_mask || (mq & MI_OOW)) { push(s, t, li, mi_src(q - s.lo, mq)); wrote_rec = true; }
      }
    }
    // pass 4: the Acks' payloads, pulled by the prober itself
#pragma unroll
    for (int q = 0; q < CH; ++q) {
      const int p = c0 + q;
      if (!ack_ok[q]) continue;
      if (SH && !is_local(s, picks[p])) {
        // a remote target's queue comes from the replicas: translated here, or handed to my own ingest
        const uint32_t qc = qrem[q];
        if (qc & Q_PBN) {
          payloads++; rumors += qc & Q_PBN;
          if (!pull_remote(picks[p], qc, tk2[q].x, &ackacc)) emit_order(i, picks[p]);
        }
        continue;
      }
      const uint32_t pj = mi_pbn(pinfo[p]);
      if (pj) {
        ackacc |= tk2[q].x;
        if (!use_mask || (pinfo[p] & MI_OOW)) { s.ackfrom[(size_t)li * s.P + nack] = mi_src(picks[p] - s.lo, pinfo[p]); nack++; wrote_rec = true; }
        payloads++; rumors += pj;
      }
    }
#pragma unroll
    for (int q = 0; q < CH; ++q) if ((uint32_t)(c0 + q) < np && valid[c0 + q] && !ack_ok[q]) failmask |= 1u << (c0 + q);
    }
    SECT(35);                                       // Acks
  }
  // pass 5 (rare without loss): probes without an Ack -> K indirect probes -> maybe Suspect (src/Core.hs:247-254).
  // The WAVE does it: the K proxies of a failed probe are independent chains i -> q -> j -> q -> i of up to four
  // deliveries each (an atomic with a returned position per delivery in a tick of explicit records), and a lane walking
  // the chains of its failed probes one after the other kept its whole wave waiting -- with 1 % loss 98 % of the waves
  // hold a failed probe and this pass was 48 % of the kernel's wave time (profiles/r03z_*).  Per probe index p: the
  // lanes whose probe p failed draw their proxies (kRandomMembers, by the prober: the draws depend on each other), the
  // (prober, proxy) pairs are dealt to the wave's lanes through LDS, every lane walks ONE chain, and the outcomes come
  // back by LDS atomics -- "any proxy relayed an Ack" is an OR over the K lanes (the aggregation across k lanes of
  // SURVEY D9).
  {
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    // "dst merges src's start-of-tick queue" on behalf of prober pi (global id; local index pli, queue mask pmask):
    // any dst / src (global ids); msrc = minfo[src]; what the prober itself pulls is ORed into *got
    auto deliver_for = [&](uint32_t pi, uint32_t pli, unsigned long long pmask, uint32_t dst, uint32_t src, uint32_t msrc,
                           unsigned long long* got) {
      const bool src_here = !SH || is_local(s, src);
      const uint32_t qsrc = src_here ? 0u : (uint32_t)s.q_all[src];   // a remote member's queue: its replicated byte
      const uint32_t cnt = src_here ? mi_pbn(msrc) : (qsrc & Q_PBN);
      if (!cnt) return;                                             // empty payload
      payloads++; rumors += cnt;
      if (!src_here) {
        if (dst == pi && pull_remote(src, qsrc, (qsrc & Q_OOW) ? 0ull : s.mask_all[src], got)) return;       // pulled by the prober itself
        emit_order(dst, src);                                       // dst's owner (maybe this shard) delivers it
        return;
      }
      if (dst == pi) {
        // pulled by the prober itself: no atomics on the mask path, (rarely) an explicit record of its own
        if (use_mask) *got |= src == pi ? pmask : s.pk[src - s.lo].x;
        if (!use_mask || (msrc & MI_OOW)) { push(s, t, pli, mi_src(src - s.lo, msrc)); wrote_rec = true; }
      } else if (!SH || is_local(s, dst)) {
        deliver_local(s, t, use_mask, stale, dst - s.lo, src - s.lo, msrc, src == pi ? pmask : (use_mask ? s.pk[src - s.lo].x : 0ull));
        wrote_rec |= !use_mask || (msrc & MI_OOW);
      } else {
        emit_order(dst, src);
      }
    };
    if (__ballot(failmask != 0u)) {
      for (int p = 0; p < PMAX; ++p) {
        const bool mine = ((failmask >> p) & 1u) != 0u;
        if (!__ballot(mine)) continue;                             // wave-uniform
        // stage A, by the probers: kRandomMembers store (numToGossip cfg) [] for proxies (src/Core.hs:249), D7: not the target
        uint32_t qs[PMAX], qinfo[PMAX];
        uint32_t nq = 0, j = 0, mj = 0;
#pragma unroll
        for (int k = 0; k < PMAX; ++k) { qs[k] = 0; qinfo[k] = 0; }
        if (mine) {
          j = picks[0]; mj = pinfo[0];
#pragma unroll
          for (int e = 1; e < PMAX; ++e) if (p == e) { j = picks[e]; mj = pinfo[e]; }
          dfail++;
          const uint32_t excl = j;
          nq = select_members<PMAX>(s, mk, i, s.K, P_PROXY, (uint32_t)p, &excl, 1, qs, qinfo, use_mask);
          if (crashes.n) {
#pragma unroll
            for (int k = 0; k < PMAX; ++k) qinfo[k] = overlay(qs[k], qinfo[k]);
          }
          preqs += nq;
          px_ctx[wv][lane] = make_uint4(mi, j, mj, 0u);
          px_mask[wv][lane] = mymask;
          px_ack[wv][lane] = 0u;
          px_got[wv][lane] = 0ull;
        }
        // stage B: the chains, PX_KG proxy indices at a time (at most 64 x PX_KG pairs in the wave's table)
        for (uint32_t k0 = 0; k0 < (uint32_t)PMAX; k0 += PX_KG) {
          const uint32_t nmine = (mine && nq > k0) ? min((uint32_t)PX_KG, nq - k0) : 0u;
          const uint32_t incl = wave_prefix_incl(nmine);
          const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
          if (!total) break;                                       // wave-uniform: nobody has a proxy index >= k0
#pragma unroll
          for (int kk = 0; kk < PX_KG; ++kk) {
            if ((uint32_t)kk < nmine) {
              uint32_t q = qs[0], mq = qinfo[0];
#pragma unroll
              for (int e = 1; e < PMAX; ++e) if (k0 + (uint32_t)kk == (uint32_t)e) { q = qs[e]; mq = qinfo[e]; }
              px_item[wv][incl - nmine + (uint32_t)kk] = make_uint4(q, mq, lane | ((k0 + (uint32_t)kk) << 8), 0u);
            }
          }
          lds_wave_sync();
          for (uint32_t x = lane; x < total; x += 64u) {
            const uint4 it = px_item[wv][x];
            const uint32_t q = it.x, mq = it.y, src = it.z & 63u, k = it.z >> 8;
            const uint4 cx = px_ctx[wv][src];
            const unsigned long long pmask = px_mask[wv][src];
            const uint32_t pli = (li - lane) + src, pi = s.lo + pli, pmi = cx.x, pj = cx.y, pmj = cx.z;
            const uint32_t idx = ((uint32_t)p << 8) | k;
            unsigned long long got = 0;
            bool ok = false;
            do {
              // i -> q : IndirectPing (src/Core.hs:250, 262-269)
              if (lost(s, tk, P_L_REQ, pi, q, idx) || !mi_up(mq)) break;
              deliver_for(pi, pli, pmask, q, pi, pmi, &got);
              // q -> j : Ping on behalf of i (src/Core.hs:105-108; D8, D12)
              if (!mi_up(pmj) || lost(s, tk, P_L_FWD, q, pj, idx)) break;
              deliver_for(pi, pli, pmask, pj, q, mq, &got);
              // j -> q : Ack
              if (lost(s, tk, P_L_BACK, pj, q, idx)) break;
              deliver_for(pi, pli, pmask, q, pj, pmj, &got);
              // q -> i : relayed Ack (D9)
              if (lost(s, tk, P_L_RELAY, q, pi, idx)) break;
              deliver_for(pi, pli, pmask, pi, q, mq, &got);
              ok = true;
            } while (false);
            if (got) atomicOr(&px_got[wv][src], got);
            if (ok) atomicOr(&px_ack[wv][src], 1u);
          }
          lds_wave_sync();
        }
        if (mine) {
          ackacc |= px_got[wv][lane];
          if (!px_ack[wv][lane]) {                                 // second unlessAck (src/Core.hs:251)
            // suspectNode store (Suspect (memberIncarnation m) name)  (src/Core.hs:253): lands in merge
            ensure_slot(s, j);
            s.fail[(size_t)li * s.P + nfail] = j;
            nfail++;
            susp++;
            if (mi_up(mj)) fsusp++;
            else atomicMin(&s.first_suspect[j], t);
          }
        }
      }
    }
  }
  SECT(36);                                         // indirect probes
  if (act) {
    s.probe_out[li] = (uint16_t)(n_pings | (nfail << 5) | (nack << 10));
  }
  if (li < s.N && !ABL(ABL_ACKMASK_STORE)) s.ackmask[li] = ackacc;
  if (__ballot(wrote_rec) && (threadIdx.x & 63u) == 0u) s.g[G_ANYREC] = t + 1u;   // one plain store per wave that wrote any (tagged with the tick: nobody has to reset it)
  ctr_add_wave(&sh, C_PINGS, n_pings);
  ctr_add_wave(&sh, C_ACTIVE, act ? 1u : 0u);
  ctr_add_wave(&sh, C_PAYLOADS, payloads);
  ctr_add_wave(&sh, C_RUMORS_SEEN, rumors);
  // the rare ones: a wave-uniform test first
  if (__ballot((dfail | susp) != 0u)) {
    ctr_add_wave(&sh, C_DIRECT_FAILED, dfail);
    ctr_add_wave(&sh, C_PING_REQS, preqs);
    ctr_add_wave(&sh, C_SUSPECTS, susp);
    ctr_add_wave(&sh, C_FALSE_SUSPECTS, fsusp);
  }
  SECT(37);                                         // outputs, counters
  ctr_flush(s, &sh, blockIdx.x);
  SECT(38);
  if (SH) {
    // The block routes its own records into the per-owner segments of q_send.  The common ones -- the Pings' payloads for remote
    // targets, {target, me} -- never left the registers (picks, pingmask): per owner a prefix sum over the wave, the waves'
    // totals through LDS, ONE reservation per owner and block, and every lane writes its records in place, neighbours next to
    // each other.  (Each of them appended to a list with an LDS atomic and read back cost the 524 288-member kernel a quarter
    // of its time: the compiler turns a per-lane atomic on one LDS word into a loop over the lanes.)
    const uint32_t lane_ = threadIdx.x & 63u, wv_ = threadIdx.x >> 6;
    auto mine_for = [&](uint32_t g) -> uint32_t {
      uint32_t c = 0;
#pragma unroll
      for (int p = 0; p < PMAX; ++p) c += (((pingmask >> p) & 1u) && (uint32_t)((pingown >> (4 * p)) & 15u) == g) ? 1u : 0u;
      return c;
    };
    for (uint32_t g = 0; g < s.n_shards; ++g) {
      const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)wave_prefix_incl(mine_for(g)), 63);
      if (lane_ == 0u) rt_wave[g][wv_] = tot;
    }
    __syncthreads();
    if (threadIdx.x < s.n_shards) {
      uint32_t acc = 0;
      for (int w = 0; w < BLOCK / 64; ++w) { const uint32_t c = rt_wave[threadIdx.x][w]; rt_wave[threadIdx.x][w] = acc; acc += c; }
      rt_base[threadIdx.x] = acc ? atomicAdd(&s.send_cnt[MAX_SHARDS + threadIdx.x], acc) : 0u;
    }
    __syncthreads();
    for (uint32_t g = 0; g < s.n_shards; ++g) {
      const uint32_t c = mine_for(g);
      const uint32_t incl = wave_prefix_incl(c);               // (wave-uniform call: every lane takes part)
      if (!c) continue;
      uint32_t pos = rt_base[g] + rt_wave[g][wv_] + incl - c;
#pragma unroll
      for (int p = 0; p < PMAX; ++p)
        if (((pingmask >> p) & 1u) && (uint32_t)((pingown >> (4 * p)) & 15u) == g) {
          if (pos < s.p_cap) s.q_send[(size_t)g * s.p_cap + pos] = make_uint2(picks[p], i);
          else atomicOr(&s.g[G_ERR], (uint32_t)ERRF_XCHG);
          pos++;
        }
    }
    // the rare ones (ord: a remote source whose payload needs more than a translation, the hops of an indirect probe): the
    // list is this block's own (written above, visible behind the barriers): count (LDS), reserve, write
    if (threadIdx.x < (uint32_t)MAX_SHARDS) rt_cnt[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t n = ordn < s.ord_cap ? ordn : s.ord_cap;
    if (n == 0u) return;                                        // (block-uniform)
    const uint2* list = s.ord + (size_t)blockIdx.x * s.ord_cap;
    for (uint32_t k = threadIdx.x; k < n; k += BLOCK) atomicAdd(&rt_cnt[owner_of(s, list[k].x)], 1u);
    __syncthreads();
    if (threadIdx.x < s.n_shards) {
      const uint32_t c = rt_cnt[threadIdx.x];
      rt_base[threadIdx.x] = c ? atomicAdd(&s.send_cnt[MAX_SHARDS + threadIdx.x], c) : 0u;
      rt_cnt[threadIdx.x] = 0;
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < n; k += BLOCK) {
      const uint2 o = list[k];
      const uint32_t peer = owner_of(s, o.x), pos = rt_base[peer] + atomicAdd(&rt_cnt[peer], 1u);
      if (pos < s.p_cap) s.q_send[(size_t)peer * s.p_cap + pos] = o;
      else atomicOr(&s.g[G_ERR], (uint32_t)ERRF_XCHG);
    }
  }
}

// ================================================================================================
// merge kernel
// ================================================================================================

// a fresh rumour id: its allocation number, whose low RID_BITS are the id (RID_PARKED is never handed out)
__device__ inline uint32_t new_rid(const DevState& s) {
  uint32_t c;
  do c = atomicAdd(&s.g[G_NRUM], 1u); while ((c & RID_MASK) == RID_PARKED);
  return c;
}

// (slot, key) -> rumour id, created by whoever states the rumour first (own probe, own timer,
// refutation, join); everyone else learns the id from the piggyback entry that carries the rumour.
// One way per (incarnation, state) combination, newer combinations evict older ones.  Duplicate ids
// for one rumour are harmless (an id is only a filter key), so every failure path just takes a fresh id.
__device__ inline uint32_t find_rid(const DevState& s, uint32_t slot, uint32_t key, uint32_t* number = nullptr) {
  // *number: the id's allocation number modulo 2^(RID_BITS + 15) (its age relative to a head: young_rid)
  const uint32_t way = ((key >> 2) * 3u + (key & 3u)) & (uint32_t)(RT_WAYS - 1);
  unsigned long long* p = s.rtab + (size_t)slot * RT_WAYS + way;
  const unsigned long long claim = (unsigned long long)(key + 1u) << 32;
  unsigned long long e = *p;                        // cached load: a published entry never reverts
  for (int spin = 0; spin < 64; ++spin) {           // a racing creator may sit in my own wave: do not wait for it long
    const uint32_t ek = (uint32_t)(e >> 32);
    if (ek == key + 1u) {
      if (e & RT_READY) {
        // a published id is reused only while it still names this rumour and is younger than half the id space:
        // the id counter wraps (RID_BITS), and an id that another rumour took over -- or one so old that it
        // would read as an id just above the head -- would set a foreign bit in somebody's mask.  The age is
        // taken from the allocation number kept with the id, not from the id: an id exactly one turn of the id
        // space old is being handed out again in this very tick, rum[] may not show it yet, and its age modulo
        // 2^RID_BITS reads as zero (found by a soak of the 8-bit build).  A stale entry is re-claimed like a free one.
        const uint32_t rid = (uint32_t)e & RID_MASK;
        const uint32_t born = ((((uint32_t)e >> RT_GEN_SHIFT) & RT_GEN_MASK) << RID_BITS) | rid;
        const uint2 r = s.rum[rid];
        if (r.x == slot && r.y == key && ((s.g[G_NRUM] - born) & RT_SPAN_MASK) < RID_FAR) { if (number) *number = born; return rid; }
      } else {
        e = atomicCAS(p, 0ull, 0ull);               // being published by another lane: re-read at device scope
        continue;
      }
    } else if (ek > key + 1u) break;                // the way belongs to a newer rumour about this subject
    const unsigned long long seen = atomicCAS(p, e, claim);
    if (seen == e) {
      const uint32_t c = new_rid(s), rid = c & RID_MASK;
      s.rum[rid] = make_uint2(slot, key);           // read by other members from the next launch on
      atomicExch(p, claim | RT_READY | ((unsigned long long)((c >> RID_BITS) & RT_GEN_MASK) << RT_GEN_SHIFT) | rid);
      if (number) *number = c & RT_SPAN_MASK;
      return rid;
    }
    e = seen;
  }
  const uint32_t c2 = new_rid(s), rid = c2 & RID_MASK;
  s.rum[rid] = make_uint2(slot, key);
  if (number) *number = c2 & RT_SPAN_MASK;
  return rid;
}

// An id handed out DURING a tick (xlat_kernel / ingest_kernel of a sharded cluster: rumours a peer knows and this shard
// does not) is read in the same tick, through foreign lines, against rings whose head is the START of the tick: it
// must not lie a whole turn of the id space minus the ring ahead of that head, or it reads as an OLD id inside the
// ring (found by a soak of the 10-bit build: 3 000 members, 20 % loss, two shards -- deliveries filtered as "known";
// the product's 16-bit ids reach that point at ~65 000 new rumours per shard and tick: heavy loss at a million
// members).  Past that point entries travel without an id (exact: every filter is skipped).  `number` = the id's
// allocation number as find_rid reports it.
__device__ inline uint32_t young_rid(uint32_t rid, uint32_t number, uint32_t H) {
  const uint32_t ahead = (number - H) & RT_SPAN_MASK;           // an id older than the head reads as a huge distance
  // (strictly below: the id H + 2^RID_BITS - KW_BITS has the low bits of H - KW_BITS, the OLDEST id of the wide window -- two rumours
  // under one id, and a receiver's test-and-set dropped the second.  Round 6, soak case 832/8 on the 8-bit build: a shard handing out
  // 192 ids in one tick; the product's 16-bit ids get there at 65 280)
  return (ahead < RID_MASK + 1u - KW_BITS || ahead > RT_SPAN_MASK / 2u) ? rid : RID_PARKED;
}

// a rumour id that fell out of the (wide) known-ring window is replaced by RID_PARKED ("no id") at the next rewrite of
// the line, so that a long-lived entry can never alias into a later window
__device__ inline uint32_t park_rid(uint32_t lo, uint32_t H) {
  const uint32_t above = (pe_rid(lo) - (H - KW_BITS)) & RID_MASK;     // distance above the window bottom
  return (pe_rid(lo) != RID_PARKED && above < KW_BITS + RID_NEAR) ? lo : pe_lo(pe_slot(lo), RID_PARKED);
}


// ================================================================================================
// records phase (the start of merge_kernel in a tick with explicit records)
// ================================================================================================
// Explicit delivery records "dst merges src's 64-B line" (swim_device.h): the exact path behind the masks -- queues
// with entries outside the mask window, every delivery of a tick that follows a burst of rumour ids (message loss),
// payloads from other shards my masks cannot carry.  A member with records reads its sources' lines, filters the
// entries through the wide known-ring (kw: "my view dominates rumour id r" for the last KW_BITS ids) and leaves the
// survivors as a todo list for the rest of the kernel: {slot | rid << 16, key}, 8 bytes each, in
// a region of `todo` reserved per wave (upper bound: 8 entries per source; one atomic per wave).  It follows
// src/Core.hs:110-117 for messages that arrive as whole Envelopes.
// Why a phase of its own, in front of everything else (measured, profiles/r03d_*, r03g_*, r03i_*): walked in the
// middle of the kernel -- with the tick's group, deadlines and counters live, and its own batch of view cells, bases
// and subjects in registers -- this path set the register allocation of the whole kernel: 128 VGPRs + 56 bytes of
// scratch against 95 and none, although a lossless tick never enters it (merge 158 -> 147 us at a million members).
// As a kernel of its own it costs 10 us per tick even when it leaves at once: every kernel boundary on this chip
// writes the XCDs' L2s back and invalidates them (5-6 us gaps between the tick's kernels in the trace).  Here it
// needs 62 registers with nothing else live, and the view cells are the todo loop's business, four at a time.
// reserve n todo entries in the workgroup's region -- or, when that is full, in the spill area all regions share (a
// degraded cluster under heavy loss sends everything to the few members it still holds Alive: one workgroup's members then
// take many times their share; found by the GPU sweep at 4 096 members, P = K = 10, 30 % loss); NONE32 = no room (loud: ERRF_TODO)
__device__ inline uint32_t todo_reserve(const DevState& s, uint32_t n) {
  const uint32_t region = blockIdx.x & (TODO_REGIONS - 1u);
  const uint32_t got = atomicAdd(&s.todo_n[region * 16u], n);
  if (got + n <= s.todo_cap && got + n >= got) return region * s.todo_cap + got;
  const uint32_t got2 = atomicAdd(&s.todo_n[TODO_REGIONS * 16u], n);
  if (got2 + n > s.todo_spill || got2 + n < got2) { atomicOr(&s.g[G_ERR], (uint32_t)ERRF_TODO); return NONE32; }
  return s.todo_spill_at + got2;
}
// the member's wide known-ring as the records meet it: what it learnt since it was written forgotten position-wise, what
// the 64-position ring knows copied in (its ids own one or two of the wide ring's words)
__device__ inline Ring256 load_wide_ring(const DevState& s, uint32_t li, uint32_t H, unsigned long long kn) {
  Ring256 kw;
  const uint32_t kwh0 = s.kw_head[li];
  const ulonglong4 v = s.kw[li];
  kw.w0 = v.x; kw.w1 = v.y; kw.w2 = v.z; kw.w3 = v.w;
  r256_forget(kw, kwh0, H - kwh0);
  const unsigned long long young = low_bits((int)(H & 63u));   // positions of the ids in H's own block of 64
  r256_or(kw, (H >> 6) & (KW_BITS / 64u - 1u), kn & young);
  r256_or(kw, ((H >> 6) - 1u) & (KW_BITS / 64u - 1u), kn & ~young);
  return kw;
}
// One member's sources walked by its own thread, survivors stored as they are found: the form for the exception (a
// lossless tick with a few out-of-window queues: the start of merge_kernel) and for what records_kernel's flattened pass
// leaves out (overflow-list sources, a member with more sources than a chunk).  first / count select the sources:
// [first, first + count) of the sequence ackfrom[0..nack), inbox[0..nin), my entries of the overflow list.
// Returns the number of survivors stored at out[0..).
__device__ inline uint32_t records_serial(const DevState& s, uint32_t t, uint32_t li, uint32_t H, bool ids_untrusted, Ring256& kw,
                                          unsigned long long& learnt, uint32_t nack, uint32_t nin, uint32_t novf, uint32_t first,
                                          uint2* out SECT_PARAM) {
  uint32_t nout = 0;
  auto source_word = [&](uint32_t x) -> uint32_t {
    if (x < nack) return s.ackfrom[(size_t)li * s.P + x];
    if (x < nack + nin) return s.inbox[(size_t)li * s.inbox_cap + (x - nack)];
    if (x < nack + nin + novf) {
      const uint2 o = s.ovf[(size_t)(t & 1u) * s.ovf_cap + (x - nack - nin)];
      if (o.x == li) return o.y;
    }
    return NONE32;
  };
  uint32_t srcw_next = source_word(first);         // one source ahead: its load travels with this source's line
  SECT(11);                                         // records: counts, region, rings
  for (uint32_t x = first; x < nack + nin + novf; ++x) {
    PSTAT(12);
    const uint32_t srcw = srcw_next;
    srcw_next = source_word(x + 1u);
    if (srcw == NONE32) continue;
    PSTAT(13);
    const uint4* line = (srcw & SRC_FOREIGN) ? s.fl + (size_t)(srcw & (SRC_FOREIGN - 1u)) * 4
                                             : line_ptr(s, srcw >> 31, srcw & 0x7FFFFFFFu);
    // the whole line in one round of loads; its entries filtered by the rings (no memory); the survivors stored as they
    // are: row bases, subjects and view cells are the todo loop's business
    uint4 ln[PB_SLOTS / 2];
#pragma unroll
    for (int h = 0; h < PB_SLOTS / 2; ++h) ln[h] = line[h];
#ifdef SWIM_SECTION_CLOCKS
    if (ln[0].x == 0xFFFFFFF1u && ln[3].w == 0xFFFFFFF3u) learnt |= 1ull;   // (measurement build: the mark below waits for the line)
#endif
    SECT(12);                                       // records: source word + line arrived
#pragma unroll
    for (int q = 0; q < PB_SLOTS; ++q) {
      const uint4 v = ln[q >> 1];
      const uint32_t lo = (q & 1) ? v.z : v.x, hi = (q & 1) ? v.w : v.y;
      bool want = pe_tx(hi) != 0u;
      const uint32_t rid = pe_rid(lo);
      // after a tick with more new ids than the width tolerates (G_RIDS_OFF) the lines written in that tick hold
      // ids that may be a whole turn of the id space apart -- two rumours under one id, or an id that reads as
      // one of this tick's window: no id of a line is trusted in this tick (found by a soak of the 8-bit build:
      // 3 000 members at 20 % loss allocate 660 ids in the first tick)
      if (want && !ids_untrusted && rid_in_wide(rid, H)) {
        if (r256_test(kw, rid)) want = false;      // view already dominates it
        else { r256_set(kw, rid); if (rid_in_ring(rid, H)) learnt |= rid_bit(rid); }
      }
      if (want) out[nout++] = make_uint2(lo, pe_key(hi));
    }
    SECT(13);                                       // records: ring filter + survivors stored
  }
  return nout;
}

__device__ __forceinline__ void records_phase(const DevState& s, uint32_t t, uint32_t li, uint32_t mi, unsigned long long got, uint32_t H, uint32_t Hprev SECT_PARAM) {
  const bool up = mi_up(mi);
  uint32_t cnt = 0, nack = 0;
  if (up) { cnt = s.inbox_cnt[li]; nack = s.probe_out[li] >> 10; }
  const bool has = (cnt | nack) != 0u;
  if (!__ballot(has)) return;
#ifdef SWIM_REC_STATS
  { const unsigned long long b = __ballot(has); if ((threadIdx.x & 63u) == 0u) { atomicAdd(&s.g[90], (uint32_t)__popcll(b)); atomicAdd(&s.g[91], 1u); } }
#endif
  const uint32_t nin = cnt < s.inbox_cap ? cnt : s.inbox_cap;
  const uint32_t novf = cnt > s.inbox_cap ? min(s.g[G_OVF0 + (t & 1u)], s.ovf_cap) : 0u;
  // room for 8 survivors per source; the overflow list is shared: count my entries first
  uint32_t mine_ovf = 0;
  for (uint32_t x = 0; x < novf; ++x) mine_ovf += s.ovf[(size_t)(t & 1u) * s.ovf_cap + x].x == li ? 1u : 0u;
  const uint32_t ub = has ? (nack + nin + mine_ovf) * (uint32_t)PB_SLOTS : 0u;
  const uint32_t incl = wave_prefix_incl(ub);
  const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
  uint32_t base = 0;
  if ((threadIdx.x & 63u) == 0u) base = todo_reserve(s, total);   // one atomic per wave
  base = (uint32_t)__builtin_amdgcn_readlane((int)base, 0);
  if (base == NONE32) { if (has) s.inbox_cnt[li] = 0; return; }    // loud (ERRF_TODO), never a silent drop
  if (!has) return;
  const uint32_t off = base + incl - ub;
  // my 64-position ring as the rest of the kernel will hold it when it comes to the records: last tick's new ids
  // forgotten, this tick's mask deliveries (got: pushed to me and pulled by me) learnt
  const unsigned long long kn = (s.pk[li].y & ~stale_positions(Hprev, H)) | got;
  unsigned long long learnt = 0;                   // ring positions of the ids the records carry: ORed into the ring later
  Ring256 kw = load_wide_ring(s, li, H, kn);
  const uint32_t nout = records_serial(s, t, li, H, s.g[G_RIDS_OFF] != 0u, kw, learnt, nack, nin, novf, 0u, s.todo + off SECT_ARG);
  s.kw[li] = make_ulonglong4(kw.w0, kw.w1, kw.w2, kw.w3);
  s.kw_head[li] = H;
  s.kn_rec[li] = learnt;
  s.todo_seg[li] = make_uint4(off, nout, 0u, 0u);
  s.inbox_cnt[li] = nout;                          // the todo loop: entries of my list (the end of the kernel clears the count)
}

// ================================================================================================
// records kernel (between probe_kernel and merge_kernel, for handles whose every tick has records: message loss, shards)
// ================================================================================================
// The same job as records_phase -- sources' lines -> ring filter -> todo lists -- when EVERY member has six or more
// sources in every tick.  What the one-thread-per-member form costs there was measured (profiles/r03t_*, 1 % loss, a
// million members: the phase is 54 % of merge_kernel's wave time): a wave walks max-over-its-lanes sources (13-14 rounds
// for a mean of 6.7); every survivor is an 8-byte scattered store, and on gfx9 a wave that waits for its next load also
// waits for every store it has in flight (one counter, in order); every wave of the grid bumps one counter.  Here:
//   * a workgroup's (member, source) pairs are dealt to its 256 threads round-robin -- every lane loads a line in every
//     round, ceil(pairs / 256) rounds;
//   * the members' wide rings live in LDS for the pass (test and set = one LDS atomic: two lanes that meet the same new
//     rumour in lines of one member keep one copy);
//   * a chunk = the pairs of whole members, at most REC_UNROLL per thread: the lines stay in registers while the
//     survivors are counted (LDS), the chunk's output is laid out member by member, reserved with one atomic and
//     written in one burst -- no load waits behind a store;
//   * overflow-list sources and a member with more sources than a chunk holds are walked by the member's own thread
//     afterwards (records_serial) into a second segment of its list.
// The order of a list's entries differs from the serial form's; nothing observable depends on it (the merge is
// commutative; duplicates are looked at twice and accepted once).
constexpr uint32_t REC_UNROLL = 2, REC_CHUNK = BLOCK * REC_UNROLL;   // pairs per thread / per workgroup in a chunk
// exclusive prefix sum over the threads of the workgroup (every thread calls); *total = the sum
__device__ inline uint32_t block_prefix_excl(uint32_t x, uint32_t* wsum /* LDS, BLOCK / 64 + 1 words */, uint32_t* total) {
  const uint32_t incl = wave_prefix_incl(x);
  __syncthreads();                                  // the previous use of wsum is over
  if ((threadIdx.x & 63u) == 63u) wsum[threadIdx.x >> 6] = incl;
  __syncthreads();
  uint32_t before = 0, all = 0;
#pragma unroll
  for (int w = 0; w < BLOCK / 64; ++w) { const uint32_t v = wsum[w]; all += v; before += (uint32_t)w < (threadIdx.x >> 6) ? v : 0u; }
  *total = all;
  return before + incl - x;
}
__global__ __launch_bounds__(BLOCK, 5) void records_kernel(DevState s, uint32_t t) {
  if (s.g[G_ANYREC] != t + 1u) return;             // uniform: nobody wrote a record this tick
  __shared__ unsigned long long kw_sh[4][BLOCK];   // the members' wide rings
  __shared__ unsigned long long learnt_sh[BLOCK];
  __shared__ uint32_t pref[BLOCK + 1];             // exclusive prefix of the members' dealt sources (acks + inbox)
  __shared__ uint32_t nack_sh[BLOCK];
  __shared__ uint32_t cntc[BLOCK], offc[BLOCK];    // a chunk's survivors per member, and where each member's run starts
  __shared__ uint32_t wsum[BLOCK / 64 + 1];
  __shared__ uint32_t chunk_base;
  const uint32_t tid = threadIdx.x, li = blockIdx.x * BLOCK + tid, i = s.lo + li;
  SECT_BEGIN(48);
  const uint32_t mi = li < s.N ? s.minfo[i] : 0u;
  const bool up = mi_up(mi);
  uint32_t cnt = 0, nack = 0;
  if (up) { cnt = s.inbox_cnt[li]; nack = s.probe_out[li] >> 10; }
  const bool has = (cnt | nack) != 0u;
  const uint32_t H = s.g[G_HEAD];
  const bool ids_untrusted = s.g[G_RIDS_OFF] != 0u;
  const uint32_t nin = cnt < s.inbox_cap ? cnt : s.inbox_cap;
  const uint32_t novf = cnt > s.inbox_cap ? min(s.g[G_OVF0 + (t & 1u)], s.ovf_cap) : 0u;
  const bool dealt = has && nack + nin <= REC_CHUNK;   // else: all of it in the serial pass
  if (has) {
    const unsigned long long got = s.inmask[li] | s.ackmask[li];
    const unsigned long long kn = (s.pk[li].y & ~stale_positions(s.g[G_PREV], H)) | got;
    const Ring256 kw = load_wide_ring(s, li, H, kn);
    kw_sh[0][tid] = kw.w0; kw_sh[1][tid] = kw.w1; kw_sh[2][tid] = kw.w2; kw_sh[3][tid] = kw.w3;
  }
  learnt_sh[tid] = 0ull; nack_sh[tid] = nack;
  uint32_t T = 0;
  const uint32_t myp = block_prefix_excl(dealt ? nack + nin : 0u, wsum, &T);
  pref[tid] = myp;
  if (tid == 0) pref[BLOCK] = T;
  uint4 seg = make_uint4(0u, 0u, 0u, 0u);
  __syncthreads();
  SECT(48);                                         // counts, rings -> LDS, prefix
  for (uint32_t m0 = 0; m0 < (uint32_t)BLOCK && pref[m0] < T;) {       // uniform: chunks of whole members
    uint32_t lo = m0 + 1u, hi = BLOCK;               // m1 = the largest member bound with at most REC_CHUNK pairs from m0 on
    while (lo < hi) { const uint32_t mid = (lo + hi + 1u) >> 1; if (pref[mid] - pref[m0] <= REC_CHUNK) lo = mid; else hi = mid - 1u; }
    const uint32_t m1 = lo, x0 = pref[m0], x1 = pref[m1];
    cntc[tid] = 0;
    __syncthreads();
    // A chunk is at most REC_UNROLL pairs per thread: their source words, then their lines, travel together and the
    // lines stay in registers until the survivors' places are known (the pass waits on memory, not on arithmetic).
    uint32_t mm[REC_UNROLL], sw[REC_UNROLL], keep[REC_UNROLL], pos0[REC_UNROLL];
    uint4 ln[REC_UNROLL][PB_SLOTS / 2];
#pragma unroll
    for (int u = 0; u < (int)REC_UNROLL; ++u) {
      const uint32_t x = x0 + tid + (uint32_t)u * BLOCK;
      mm[u] = NONE32; sw[u] = 0u; keep[u] = 0u; pos0[u] = 0u;
      if (x < x1) {
        uint32_t a = m0, b = m1;                     // the member whose pair x is: the last one with pref <= x
        while (b - a > 1u) { const uint32_t mid = (a + b) >> 1; if (pref[mid] <= x) a = mid; else b = mid; }
        const uint32_t k = x - pref[a], ml = blockIdx.x * BLOCK + a, na = nack_sh[a];
        mm[u] = a;
        sw[u] = k < na ? s.ackfrom[(size_t)ml * s.P + k] : s.inbox[(size_t)ml * s.inbox_cap + (k - na)];
      }
    }
#pragma unroll
    for (int u = 0; u < (int)REC_UNROLL; ++u) {
#pragma unroll
      for (int h = 0; h < PB_SLOTS / 2; ++h) ln[u][h] = make_uint4(0u, 0u, 0u, 0u);
      if (mm[u] != NONE32) {
        const uint4* line = (sw[u] & SRC_FOREIGN) ? s.fl + (size_t)(sw[u] & (SRC_FOREIGN - 1u)) * 4
                                                   : line_ptr(s, sw[u] >> 31, sw[u] & 0x7FFFFFFFu);
#pragma unroll
        for (int h = 0; h < PB_SLOTS / 2; ++h) ln[u][h] = line[h];
      }
    }
#ifdef SWIM_SECTION_CLOCKS
    if (ln[0][0].x == 0xFFFFFFF1u && ln[REC_UNROLL - 1][3].w == 0xFFFFFFF3u) atomicOr(&learnt_sh[tid], 1ull);   // (the mark below waits for the lines)
#endif
    SECT(49);                                       // pair -> member search, source words, lines
#pragma unroll
    for (int u = 0; u < (int)REC_UNROLL; ++u) {
      if (mm[u] == NONE32) continue;
      const uint32_t m = mm[u];
#pragma unroll
      for (int q = 0; q < PB_SLOTS; ++q) {
        const uint4 v = ln[u][q >> 1];
        const uint32_t elo = (q & 1) ? v.z : v.x, ehi = (q & 1) ? v.w : v.y;
        bool want = pe_tx(ehi) != 0u;
        const uint32_t rid = pe_rid(elo);
        if (want && !ids_untrusted && rid_in_wide(rid, H)) {       // (records_serial says why ids may be untrusted)
          const uint32_t qq = rid & (KW_BITS - 1u);
          const unsigned long long bit = 1ull << (qq & 63u);
          unsigned long long* w = &kw_sh[qq >> 6][m];
          if (*w & bit) want = false;                // view already dominates it
          else if (atomicOr(w, bit) & bit) want = false;   // another lane met it first in another line of this member
          else if (rid_in_ring(rid, H)) atomicOr(&learnt_sh[m], rid_bit(rid));
        }
        keep[u] |= want ? 1u << q : 0u;
      }
      if (keep[u]) pos0[u] = atomicAdd(&cntc[m], (uint32_t)__popc(keep[u]));   // this line's run inside its member's
    }
    SECT(50);                                       // ring filter (LDS atomics)
    __syncthreads();                                  // every line of the chunk is counted
    uint32_t total = 0;
    const uint32_t myo = block_prefix_excl(cntc[tid], wsum, &total);
    offc[tid] = myo;
    if (tid == 0) chunk_base = total ? todo_reserve(s, total) : 0u;
    __syncthreads();
    SECT(51);                                       // chunk barrier, layout
    const uint32_t cb = chunk_base;
    if (cb != NONE32) {
      // the survivors, member by member: a chunk's output is one compact run of the todo buffer (a few KB), written
      // in one burst with no load behind it
#pragma unroll
      for (int u = 0; u < (int)REC_UNROLL; ++u) {
        if (!keep[u]) continue;
        uint2* out = s.todo + (size_t)cb + offc[mm[u]] + pos0[u];
        uint32_t n = 0;
#pragma unroll
        for (int q = 0; q < PB_SLOTS; ++q) {
          const uint4 v = ln[u][q >> 1];
          if ((keep[u] >> q) & 1u) out[n++] = make_uint2((q & 1) ? v.z : v.x, pe_key((q & 1) ? v.w : v.y));
        }
      }
      if (tid >= m0 && tid < m1 && dealt) seg = make_uint4(cb + myo, cntc[tid], 0u, 0u);
    }
    __syncthreads();
    SECT(52);                                       // write-out
    m0 = m1;
  }
  // what the deal left out: overflow-list sources, a member with more sources than a chunk -- by its own thread
  if (has && (novf || !dealt)) {
    uint32_t mine_ovf = 0;
    for (uint32_t x = 0; x < novf; ++x) mine_ovf += s.ovf[(size_t)(t & 1u) * s.ovf_cap + x].x == li ? 1u : 0u;
    const uint32_t nsrc = (dealt ? 0u : nack + nin) + mine_ovf;
    const uint32_t base = nsrc ? todo_reserve(s, nsrc * (uint32_t)PB_SLOTS) : NONE32;
    if (base != NONE32) {
      Ring256 kw; kw.w0 = kw_sh[0][tid]; kw.w1 = kw_sh[1][tid]; kw.w2 = kw_sh[2][tid]; kw.w3 = kw_sh[3][tid];
      unsigned long long learnt = learnt_sh[tid];
      unsigned long long dummy_clock = 0; (void)dummy_clock;
      const uint32_t n1 = records_serial(s, t, li, H, ids_untrusted, kw, learnt, nack, nin, novf, dealt ? nack + nin : 0u, s.todo + base
#ifdef SWIM_SECTION_CLOCKS
                                         , dummy_clock
#endif
                                         );
      kw_sh[0][tid] = kw.w0; kw_sh[1][tid] = kw.w1; kw_sh[2][tid] = kw.w2; kw_sh[3][tid] = kw.w3;
      learnt_sh[tid] = learnt;
      seg.z = base; seg.w = n1;
    }
  }
  if (has) {
    s.kw[li] = make_ulonglong4(kw_sh[0][tid], kw_sh[1][tid], kw_sh[2][tid], kw_sh[3][tid]);
    s.kw_head[li] = H;
    s.kn_rec[li] = learnt_sh[tid];
    s.todo_seg[li] = seg;
    s.inbox_cnt[li] = seg.y + seg.w;                 // merge_kernel: entries of my list (it clears the count)
  }
  SECT(53);                                         // serial pass (overflow list, oversized members), final stores
}

// One thread = one member's end of tick:
//   suspicion timers (the FIXME at src/Core.hs:141; D4), own probes that ended without an ack
//   (src/Core.hs:253) and the rumours delivered this tick (src/Core.hs:110-117) go through the state
//   rule suspectOrDeadNode' (src/Core.hs:142-187) + the unwritten aliveNode (:197-218, D6) as the
//   commutative merge entry := max(entry, (incarnation,state)) (H3, D13), with what follows
//   `saveMember m'` (:169-179): lastChange, timer start, enqueue for piggybacking, membership event,
//   digest; then the piggyback queue `disseminate` leaves as a FIXME (src/Core.hs:136-138; D5) is rebuilt.
// Delivered rumours arrive as masks: new = (pushed | pulled) & ~known is the whole filter, and the
// lanes of a wave walk their new bits in the same order, so their view / timer accesses coalesce.
// Measured on MI355X (profiles/r02w_variants.txt): the kernel is bound by the instructions a wave executes, not
// by occupancy (3, 4 and 5 waves per SIMD within 1 %; 5 spills) -- 4 waves leave 128 registers; a batch of 2
// rumours beats 4 (fewer unrolled copies of the state rule executed by a wave whose lanes hold 1-2 new rumours)
#ifndef SWIM_GOSSIP_BATCH       // rumours whose loads are issued together in merge_kernel
#define SWIM_GOSSIP_BATCH 2
#endif
#ifndef SWIM_MERGE_WAVES         // measured with the todo batch (profiles/r03af_*; lossless / 1 % loss, us per merge): 4 waves x batch 2:
#define SWIM_MERGE_WAVES 4       // 144 / 643; 5 x 2: 147 / 628; 4 x 4: 154 / 634; 5 x 4 (spills): 166 / 647
#endif
#ifndef SWIM_MERGE_SORT         // 1: a workgroup's members are dealt to its threads in descending order of work (merge_kernel)
#define SWIM_MERGE_SORT 0
#endif
#ifndef SWIM_TODO_BATCH         // todo entries (explicit records' survivors) whose view cells merge_kernel loads together
#define SWIM_TODO_BATCH 2
#endif
#ifndef SWIM_MERGE_UNION        // 1: the delivered rumours are DECIDED in a wave-uniform walk over the union of the lanes' new ring positions --
#define SWIM_MERGE_UNION 0      //    per position a predicated load / compare / store of ONE view row for the wave's 64 consecutive members -- and
#endif                          //    booked (queue, deadlines, digest, events) by a per-lane walk over what was accepted; 0 (product): each lane walks
                                //    its own positions.  Round 6, interleaved A/B on one cluster (profiles/r06b_ab_merge_union.txt): the union walk is
                                //    SLOWER -- 157.4 against 144.7 us saturated, 728 against 707 us at 1 % loss: it saves sector requests (1.4 cells of a
                                //    wave share a sector) but doubles the rounds of dependent loads a wave waits for (popc(union) / batch against the
                                //    busiest lane's count / 2), and the kernel is bound by those round trips at the chip's random-access rate.
#ifndef SWIM_RING_DIR           // 1: a rumour this member STATES itself (a suspicion deadline firing, a failed probe) is looked up in the tick's ring
#define SWIM_RING_DIR 1         //    first -- an LDS hash of the 64 positions -- and takes its id and subject from there: the Dead a deadline declares
#endif                          //    is in circulation already for all but the first members to declare it; find_rid + subject_of + minfo were four
                                //    dependent memory round trips per firing deadline (round 6)
constexpr uint32_t RDIR_SLOTS = 128;
__device__ inline uint32_t rdir_hash(uint32_t slot, uint32_t key) { return (slot * 0x9E3779B1u + key * 0x85EBCA77u) >> 25; }   // 7 bits
#ifndef SWIM_UNION_BATCH        //    positions (rounds 2-5; A/B)
#define SWIM_UNION_BATCH 4      // union positions whose view cells are loaded together
#endif
#ifndef SWIM_ACC_CAP
#define SWIM_ACC_CAP 8
#endif
constexpr int ACC_CAP = SWIM_ACC_CAP;            // accepted rumours a lane parks in LDS between two bookkeeping passes
constexpr int ASM_STRIDE = BLOCK + 2;   // words per LDS column: keeps the transposed line store conflict-free

// Settling, the per-member part (swim_device.h; begin_kernel builds the lists, settle_finish commits):
// every member -- up or down -- shows its entry of each eligible row (the maximum over the members that
// are up goes to settle_key) and clears its cell of every row settled at the end of the last tick.
__device__ inline void settle_pass(const DevState& s, uint32_t li, bool up, uint32_t* wmax /* LDS, one word per wave */) {
  const uint32_t ns = s.g[G_SETTLE_N], nz = s.g[G_ZERO_N];
  for (uint32_t k = 0; k < ns; ++k) {
    const uint32_t key = up ? v_key(s, vidx(s, li, s.settle_slots[k])) : 0u;
    const uint32_t m = wave_max(key);
    if ((threadIdx.x & 63u) == 0u) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {                       // one partial per (row, block): no same-address atomics
      uint32_t bm = 0;
      for (int w = 0; w < BLOCK / 64; ++w) bm = max(bm, wmax[w]);
      s.settle_part[(size_t)k * s.nblocks + blockIdx.x] = bm;
    }
    __syncthreads();
  }
  if (li < s.N)
    for (uint32_t k = 0; k < nz; ++k) v_put(s, vidx(s, li, s.zero_slots[k]), 0u, 0u);
}

__global__ __launch_bounds__(BLOCK, SWIM_MERGE_WAVES) void merge_kernel(SWIM_STATE_PARAM, uint32_t t, uint32_t rec_inline, uint32_t folded) {
  SWIM_STATE_BIND
  __shared__ BlockCounters sh;
  __shared__ uint32_t asm_[PB_SLOTS * 2][ASM_STRIDE];   // the outgoing line is assembled here: [2 entry + word][thread]
  __shared__ uint32_t gsubj[PB_SLOTS][ASM_STRIDE];      // subjects of this tick's group (its sort key)
  __shared__ uint32_t wfl[BLOCK];
  __shared__ uint32_t wmax[BLOCK / 64];
  __shared__ uint4 ring_sh[KN_BITS];                    // this tick's ring (begin_kernel): position -> {slot, key, base, subject}
#if SWIM_RING_DIR
  __shared__ uint32_t rdir[RDIR_SLOTS];                 // (slot, key) -> ring position + 1: the tick's ring as a hash table (below)
#endif
#if SWIM_MERGE_UNION
  __shared__ uint32_t acc_sh[ACC_CAP][ASM_STRIDE];      // accepted rumours awaiting their bookkeeping: position | changed-already << 6 | (key - old key) << 7
#endif
  const uint32_t tid = threadIdx.x;
#if SWIM_MERGE_SORT
  // Members to threads BY WORK: a wave executes the state rule as often as its busiest lane (the rumour loop runs max-over-lanes
  // batches: ~4 for a mean of 1.3 per member), so the workgroup's 256 members are dealt to its threads in descending order of
  // the rumours they have not seen yet (+ failed probes + explicit-record entries): the busiest 64 share a wave, the idle
  // ones another.  Everything below addresses the member through li; its queue line is assembled in its thread's LDS
  // columns as before.
  __shared__ uint32_t sort_cnt[16], sort_base[16];
  __shared__ uint32_t perm_sh[BLOCK];
  uint32_t li;
  {
    const uint32_t li0 = blockIdx.x * BLOCK + tid;
    uint32_t w = 0;
    if (li0 < s.N && mi_up(s.minfo[s.lo + li0])) {
      const unsigned long long f = (s.inmask[li0] | s.ackmask[li0]) & ~(s.pk[li0].y & ~stale_positions(s.g[folded ? G_PREV_NEW : G_PREV], s.g[folded ? G_HEAD_NEW : G_HEAD]));
      w = min(15u, 1u + (uint32_t)__popcll(f) + ((s.probe_out[li0] >> 5) & 31u) + min(s.inbox_cnt[li0], 6u));
    }
    if (tid < 16u) sort_cnt[tid] = 0;
    __syncthreads();
    const uint32_t r = atomicAdd(&sort_cnt[w], 1u);
    __syncthreads();
    if (tid == 0) { uint32_t b = 0; for (int k = 15; k >= 0; --k) { sort_base[k] = b; b += sort_cnt[k]; } }
    __syncthreads();
    perm_sh[sort_base[w] + r] = tid;
    __syncthreads();
    li = blockIdx.x * BLOCK + perm_sh[tid];
  }
#else
  const uint32_t li = blockIdx.x * BLOCK + threadIdx.x;
#endif
  const uint32_t i = s.lo + li;                    // global id
  SECT_BEGIN(0);
  if (blockIdx.x == 0 && threadIdx.x == 0) s.g[G_OVF0 + ((t + 1) & 1u)] = 0;  // next tick's overflow list
  const uint32_t mi = li < s.N ? s.minfo[i] : 0u;
  const bool up = mi_up(mi);
  // (`folded`: a tick without begin_kernel -- probe_kernel's workgroup 0 left the window heads in the _NEW words; my workgroup 0
  // commits them, so that whatever comes next -- a begin_kernel, a reader on the host -- finds G_HEAD / G_PREV as always.  No
  // workgroup of this launch reads those two.)
  const uint32_t H = s.g[folded ? G_HEAD_NEW : G_HEAD], Hprev = s.g[folded ? G_PREV_NEW : G_PREV];
  if (folded && blockIdx.x == 0 && threadIdx.x == 0) { s.g[G_PREV] = Hprev; s.g[G_HEAD] = H; }
  const unsigned long long stale = stale_positions(Hprev, H);

  // ---- this member's inputs of the tick (coalesced)
  uint32_t nsent = 0, nfail = 0, cnt = 0;
  unsigned long long pushed = 0, pulled = 0;
  uint2 hot0 = make_uint2(0u, 0u);
  uint4 due = make_uint4(0u, 0u, 0u, 0u);          // deadline row of this tick
  // this tick's deadline row: a wave-uniform base (scalar registers) indexed by li at both ends of the kernel --
  // a 64-bit per-lane index held from the load to the final store was spilled to scratch, and its reload waited
  // for every store the wave had in flight
  uint4* const trow_now = s.trow + (size_t)(t % s.S) * s.N;
  // ONE round of loads with the member's own word (round 6): every coalesced per-member stream the tick reads -- issued for every
  // member of the handle, up or not (nearly all are; behind `up` they were a second round trip in front of the kernel's first
  // barrier), the known-ring and the inbox count among them (they waited for the round after)
  ulonglong2 pk0 = make_ulonglong2(0ull, 0ull);
  uint32_t cnt0 = 0;
  const bool rids_off = s.g[G_RIDS_OFF] != 0u;     // (a scalar load: with the others at the start, not in front of the queue rebuild)
  if (li < s.N) {
    const uint32_t po = s.probe_out[li];
    const unsigned long long pu = s.inmask[li], pl = s.ackmask[li];
    const uint2 h0 = s.hot[li];
    const uint4 d0 = trow_now[li];
    pk0 = s.pk[li];
    cnt0 = s.inbox_cnt[li];
    if (up) { nsent = po & 31u; nfail = (po >> 5) & 31u; pushed = pu; pulled = pl; hot0 = h0; due = d0; }
  }
  // the records phase, behind the loads above (the wait for its flag -- a scalar load at the cold start of the kernel --
  // used to stand in front of them: 7 500 clocks per wave in a tick without records, profiles/r03r_*)
  const bool recs_inline = rec_inline && s.g[G_ANYREC] == t + 1u;   // wave-uniform: somebody wrote an explicit record this tick
  if (recs_inline) records_phase(s, t, li, mi, pushed | pulled, H, Hprev SECT_ARG);
  SECT(14);                                         // records phase
  if (up) cnt = recs_inline ? s.inbox_cnt[li] : cnt0;   // entries of my todo list (the phase above leaves its length there)
  if (s.G) settle_pass(s, li, up, wmax);
  uint4 ring_mine = make_uint4(NONE32, 0u, 0u, 0u);
  if (tid < KN_BITS) { ring_mine = s.ring[tid]; ring_sh[tid] = ring_mine; }
#if SWIM_RING_DIR
  if (tid < RDIR_SLOTS) rdir[tid] = 0u;
#endif
  ctr_init(&sh);                                   // its barrier also publishes the ring
#if SWIM_RING_DIR
  // the ring as a hash table keyed by (slot, key): open addressing, 64 entries in 128 slots, claimed by LDS compare-and-swap.  Two ids
  // may name one rumour (its cache way was taken over in between, find_rid): either one serves -- ids are never observable.
  // (an id handed out before its row was given to the row's present subject names a rumour of the PREVIOUS subject -- settling
  // recycles rows, and a row may have changed hands in this very tick, after the ring was built: not in the table)
  if (tid < KN_BITS && ring_mine.x < s.R_phys && (int32_t)(rid_at(tid, H) - s.slot_born[ring_mine.x]) >= 0) {
    uint32_t h = rdir_hash(ring_mine.x, ring_mine.y);
    while (atomicCAS(&rdir[h], 0u, tid + 1u) != 0u) h = (h + 1u) & (RDIR_SLOTS - 1u);
  }
  lds_barrier();
#endif
  const uint32_t pcount = mi_pbn(mi), cur = mi_buf(mi);
  const uint32_t my_slot1 = mi & MI_SLOT;          // slot+1 of rumours about me
  const bool woke = (hot0.y & 1u) != 0;            // came back up: deadlines it slept through are still in trow
  const bool timer_due = (due.x | due.y | due.z | due.w) != 0u;
  const bool act = up && ((pushed | pulled) != 0ull || (cnt | nfail | pcount | (uint32_t)timer_due | (uint32_t)woke));
  // idle this tick: only keep the ring valid (swim_device.h); nothing to write when no id was allocated
  if (up && !act && stale) { if (pk0.y & stale) s.pk[li] = make_ulonglong2(pk0.x, pk0.y & ~stale); }

  uint32_t wflag = 0;                              // bit 0: my line was rebuilt in asm_, bit 1: into which buffer
  uint32_t self_inc = hot0.x;
  unsigned long long kn = 0;                       // known-ring, positions of last tick's new ids forgotten
  // this tick's changes = the tx == L group that heads the next piggyback line, kept sorted by subject so that
  // the line's priority order (tx desc, subject asc) holds by construction (H3, D5).  It is built in place in
  // the thread's LDS columns: entry k = {asm_[2k] = slot | rid<<16, asm_[2k+1] = key}, gsubj[k] its subject.
  uint32_t gn = 0;
  uint32_t killmask = 0;                           // own entries superseded by the new group
  uint32_t oslot[PB_SLOTS / 2];                    // slot ids of the own queue, two per register
#pragma unroll
  for (int h = 0; h < PB_SLOTS / 2; ++h) oslot[h] = 0xFFFFFFFFu;   // no entry: matches no slot (< 0xFFFF)
  const uint4* own_line = reinterpret_cast<const uint4*>(s.pb + ((size_t)cur * s.N + (li < s.N ? li : 0u)) * PB_SLOTS);
  uint32_t refute = NONE32;
  unsigned changes = 0, timers_fired = 0, evdropped = 0, refutes = 0, pb_writes = 0;   // timers_fired: low half; high half = ... about a member that is up
  unsigned long long evd = 0, ha = 0;
  TimerCell tnew; tnew.lo = 0; tnew.hi = 0; tnew.n = 0;   // deadlines t + S: go to the row just consumed

  SECT(0);                                          // inputs
  // second round of loads, issued together: the known-ring, the own queue line and the view cells of the first
  // DB deadlines of this tick's cell (the kernel spends its time waiting on such round trips, one after the
  // other: profiles/r02u_section_clocks_before.txt)
  constexpr int DB = 4;
  uint32_t dsl[DB]; VCell dcell[DB];
#pragma unroll
  for (int k = 0; k < DB; ++k) { dsl[k] = 0; dcell[k] = VCell{0u, 0u}; }
  const bool plain_due = timer_due && !woke && (uint32_t)(due.w >> 16) != TR_FULL;
  // ... and (round 6) the view cells of the first GB delivered rumours the known-ring does not cover: with the ring in round one the
  // positions are known here, a round trip before the rumour loop asks for them.  A cell this thread stores to in between (a
  // deadline or a failed probe about the same subject) is looked at again (pf_ok).
  constexpr int GBP = SWIM_GOSSIP_BATCH;
  uint32_t pf_pos[GBP], pf_slot[GBP], pf_ok = 0; VCell pf_e[GBP];
#pragma unroll
  for (int k = 0; k < GBP; ++k) { pf_pos[k] = NONE32; pf_slot[k] = NONE32; pf_e[k] = VCell{0u, 0u}; }
  uint4 ol[PB_SLOTS / 2];                          // the own queue line: stays in registers for the queue rebuild
#pragma unroll
  for (int h = 0; h < PB_SLOTS / 2; ++h) ol[h] = make_uint4(0u, 0u, 0u, 0u);
  if (act) {
    PSTAT(0);
    const unsigned long long known0 = pk0.y;
    if (pcount) {
      PSTAT(1);
#pragma unroll
      for (int h = 0; h < PB_SLOTS / 2; ++h) ol[h] = ABL(ABL_OWN_LINE) ? make_uint4(0u, 0u, 0u, 0u) : own_line[h];
    }
    if (plain_due) {
#pragma unroll
      for (int k = 0; k < DB; ++k) {
        dsl[k] = tc_get(due, k);
        if (dsl[k] && !ABL(ABL_V_LOAD)) dcell[k] = v_hot(s, vidx(s, li, dsl[k] - 1));
      }
    }
    kn = known0 & ~stale;
#if !SWIM_MERGE_UNION
    {
      unsigned long long pre = (pushed | pulled) & ~kn;   // (the own queue's positions are not taken off yet: a few cells asked for in vain)
      if (ABL(ABL_RUMOURS)) pre = 0;
#pragma unroll
      for (int k = 0; k < GBP; ++k) {
        if (!pre) break;
        const uint32_t p = (uint32_t)__ffsll((unsigned long long)pre) - 1u;
        pre &= pre - 1ull;
        const uint32_t sl = ring_sh[p].x;
        if (sl + 1 == my_slot1 || sl >= s.R_phys) continue;
        pf_pos[k] = p; pf_slot[k] = sl; pf_ok |= 1u << k;
        if (!ABL(ABL_V_LOAD)) pf_e[k] = v_hot(s, vidx(s, li, sl));
      }
    }
#endif
    if (pcount) {
#pragma unroll
      for (int h = 0; h < PB_SLOTS / 2; ++h)
        oslot[h] = (pe_tx(ol[h].y) ? pe_slot(ol[h].x) : 0xFFFFu) | ((pe_tx(ol[h].w) ? pe_slot(ol[h].z) : 0xFFFFu) << 16);
#ifndef SWIM_NO_OWN_KNOWN
      // My view dominates every rumour of my own queue (I accepted or stated each of them, entries only grow): their ring
      // positions are KNOWN.  The ring learns what gossip delivers (`fresh` below) -- but a rumour this member STATED itself (its
      // suspicion deadline firing: one per member and crash) was news to the ring when the same rumour came round by gossip: one
      // futile view-cell load per member-tick of the saturated regime, a quarter of the kernel's scattered loads (round 5).
#pragma unroll
      for (int h = 0; h < PB_SLOTS / 2; ++h) {
        if (pe_tx(ol[h].y) && rid_in_ring(pe_rid(ol[h].x), H)) kn |= rid_bit(pe_rid(ol[h].x));
        if (pe_tx(ol[h].w) && rid_in_ring(pe_rid(ol[h].z), H)) kn |= rid_bit(pe_rid(ol[h].z));
      }
#endif
    }
  }
  auto kill_slot = [&](uint32_t slot) {
#pragma unroll
    for (int h = 0; h < PB_SLOTS / 2; ++h) {
      if ((oslot[h] & 0xFFFFu) == slot) killmask |= 1u << (2 * h);
      if ((oslot[h] >> 16) == slot) killmask |= 1u << (2 * h + 1);
    }
  };
  auto group_put = [&](uint32_t slot, uint32_t rid, uint32_t key, uint32_t subj) {
    uint32_t pos = 0;
    PSTAT(2);
    for (uint32_t k = 0; k < gn; ++k) {
      PSTAT(3);
      if (pe_slot(asm_[2 * k][tid]) == slot) { asm_[2 * k][tid] = pe_lo(slot, rid); asm_[2 * k + 1][tid] = key; return; }
      pos += gsubj[k][tid] < subj ? 1u : 0u;
    }
    if (pos >= (uint32_t)PB_SLOTS) return;                  // worse than the 8 kept (largest subjects drop)
    for (uint32_t k = min(gn, (uint32_t)PB_SLOTS - 1u); k > pos; --k) {
      PSTAT(4);
      asm_[2 * k][tid] = asm_[2 * k - 2][tid]; asm_[2 * k + 1][tid] = asm_[2 * k - 1][tid]; gsubj[k][tid] = gsubj[k - 1][tid];
    }
    asm_[2 * pos][tid] = pe_lo(slot, rid); asm_[2 * pos + 1][tid] = key; gsubj[pos][tid] = subj;
    if (gn < (uint32_t)PB_SLOTS) gn++;
  };
  // The state rule on one proposal (slot, key).
  // a deadline t + S for this tick's cell; the 8th spills the cell into the row's overflow pool and links it
  const uint32_t row_now = t % s.S, par_now = (t / s.S) & 1u;
  auto tput = [&](uint32_t slot1) {
    if (tnew.n >= tc_cap(tnew)) {
      if (((uint32_t)(tnew.hi >> 48)) == TR_FULL) return;            // already "look everywhere"
      SECT_COUNT(22);
      const uint32_t sub = blockIdx.x & (s.tovf_nsub - 1u);
      const uint32_t got = atomicAdd(&s.tovf_n[((row_now * 2u + par_now) * s.tovf_nsub + sub) * 16u], 1u);
      const uint32_t idx = sub * s.tovf_sub_cap + got;
      if (got < s.tovf_sub_cap) {
        s.tovf[((size_t)row_now * 2u + par_now) * s.tovf_cap + idx] = tc_pack(tnew);
        tc_clear(tnew);
        tc_set_link(tnew, idx);
      } else { SECT_COUNT(19); tnew.hi |= (unsigned long long)TR_FULL << 48; return; }
    }
    tc_set(tnew, tnew.n, slot1);
    tnew.n++;
  };
  // `have`: what the caller already holds (loaded in a batch with its neighbours, below) -- the view cell, the
  // row's base, its subject
  enum { HAVE_CELL = 1, HAVE_BASE = 2, HAVE_SUBJ = 4, EX_LOAD = 0, EX_ALL = HAVE_CELL | HAVE_BASE | HAVE_SUBJ };
  int psite = 47; (void)psite;     // PSITE/PSTAT: path statistics of the host emulation, nothing in the product
  // an ACCEPTED proposal's bookkeeping (the view cell is stored already): everything `saveMember m'` / `Just msg` entail beyond the
  // entry itself (src/Core.hs:169-179) -- digest, counters, the deadline of a new suspicion, the queue, the event.  dkey = new key
  // - old key; again_: the entry changed in this tick before.
  auto account = [&](uint32_t slot, uint32_t key, uint32_t cause, bool hasrid, uint32_t rid_in, uint32_t dkey, bool again_, uint32_t subject, bool stated) {
    if (s.G) s.slot_last[slot] = t;              // same value from every writer
    if (!ha) ha = mix64(mix64((uint64_t)TAG_EV) + (((uint64_t)t << 32) | i));
    // the running event digest moves by E(t, i, subject) * (key - old key): linear in the key, so the changes of one
    // entry in one tick telescope whatever their order -- one hash per change (three before: 6 % of the kernel)
    if (!ABL(ABL_EVD)) evd += (mix64(ha + subject) | 1ull) * (unsigned long long)dkey;
    changes += again_ ? 0u : 1u;
    // Suspect -> Dead by timeout; ... of a member that is up all the same (a false positive: ground truth, replicated)
    if (cause == 1u) timers_fired += 1u + ((uint32_t)(s.mb[subject] & MB_UP) << 16);
    if ((key & 3u) == ST_SUSPECT) tput(slot + 1);                     // deadline t + S (D4)
    uint32_t born = 0;                            // (find_rid: the id's allocation number, i.e. its exact age against the head)
    const uint32_t rid = (hasrid || ABL(ABL_FIND_RID)) ? rid_in : find_rid(s, slot, key, &born);
#ifndef SWIM_NO_OWN_KNOWN
    // a rumour I state under an id of the window [H - 64, H): known from now on.  By the id's ALLOCATION NUMBER, not by the id: an id
    // handed out in this very tick lies above the head, and past 2^RID_BITS - 64 new ids in one tick its low bits read as a position of
    // the window -- a foreign bit in the ring, and the rumour that owns the position was filtered as known (8-bit test build, soak
    // case 832/8: 1 500 members, robust scheme, 4 shards; the product's 16-bit ids get there at 65 472 new rumours in one tick).  An
    // id the ring directory named (hasrid) is in the window by construction.
    if (stated && (hasrid ? rid_in_ring(rid, H) : ((H - 1u - born) & RT_SPAN_MASK) < KN_BITS)) kn |= rid_bit(rid);
#endif
    if (!ABL(ABL_GROUP)) {
    kill_slot(slot);
    group_put(slot, rid, key, subject);          // `Just msg` -> Broadcast -> enqueue (D5)
    }
    if (s.event_mask & (1u << cause)) {
      const uint32_t pos = atomicAdd(&s.g[G_EVCUR], 1u);
      if (pos < s.event_cap) s.events[pos] = make_uint4(t, i, subject, (key << 8) | cause);
      else evdropped++;
    }
  };
  auto examine_with = [&](uint32_t slot, uint32_t key, uint32_t cause, bool hasrid, uint32_t rid_in, int have, VCell e,
                          uint32_t sbase, uint32_t subject) {
    if (slot + 1 == my_slot1) {
      // about self -> refute (src/Core.hs:155-166): remember the largest non-Alive incarnation
      if ((key & 3u) != ST_ALIVE) refute = (refute == NONE32 || (key >> 2) > refute) ? (key >> 2) : refute;
      return;
    }
    PSTAT(5); SECT_COUNT(20);
    if (!(have & HAVE_CELL) && !ABL(ABL_V_LOAD)) e = v_hot(s, vidx(s, li, slot));
    const uint32_t curk = e.key ? e.key : ((have & HAVE_BASE) ? sbase : s.slot_base[slot]);   // untouched cell: the settled base
    if (key <= curk) return;                     // old incarnation / weaker state: ignore (:151)
    // strict reference rules: the literal livenessCheck (src/Core.hs:182-184) -- a Suspect only on an Alive entry, a Dead unless the
    // entry is Dead already, whatever the incarnations (the merge above would take both at a higher incarnation: D13)
    if (s.strict && (((key & 3u) == ST_SUSPECT && (curk & 3u) != ST_ALIVE) || ((key & 3u) == ST_DEAD && (curk & 3u) == ST_DEAD))) return;
    PSTAT(6); PSTAT(psite); SECT_COUNT(21);
    const bool again_ = v_changed_in(s, vidx(s, li, slot), e, t);     // (before the store below)
    if (!ABL(ABL_V_STORE)) v_put(s, vidx(s, li, slot), key, t + 1);                    // memberLastChange = now (:176)
#pragma unroll
    for (int k = 0; k < GBP; ++k) if (pf_slot[k] == slot) pf_ok &= ~(1u << k);     // a cell asked for ahead of time is stale now
    bool stated = !hasrid;                       // a rumour I state myself: the ring learns it (below)
#if SWIM_RING_DIR
    if (!hasrid) {
      // in circulation already?  Then the tick's ring names its id and its subject: no find_rid, no subject_of (two + one
      // dependent loads; the fourth, ground truth about the subject for FALSE_DEADS, is a byte of the L2-resident mb table)
      for (uint32_t h = rdir_hash(slot, key);; h = (h + 1u) & (RDIR_SLOTS - 1u)) {
        const uint32_t v = rdir[h];
        if (!v) break;
        const uint4 r = ring_sh[v - 1u];
        if (r.x == slot && r.y == key) { hasrid = true; rid_in = rid_at(v - 1u, H) & RID_MASK; subject = r.w; have |= HAVE_SUBJ; break; }
      }
    }
#endif
    if (!(have & HAVE_SUBJ)) subject = s.subject_of[slot];
    account(slot, key, cause, hasrid, rid_in, key - curk, again_, subject, stated);
  };
  auto examine = [&](uint32_t slot, uint32_t key, uint32_t cause, bool hasrid, uint32_t rid_in) {
    examine_with(slot, key, cause, hasrid, rid_in, EX_LOAD, VCell{0u, 0u}, 0u, 0u);
  };
  // one entry of this tick's deadline cell: Suspect since t' with t' + S <= t => Dead at the same incarnation
  // (D4); a deadline still ahead that belongs to this row goes back into the cell (a fixture of
  // swimsim_set_view); anything else (refuted, already Dead, reclaimed, superseded by a later suspicion with
  // its own cell) is dropped
  auto deadline = [&](uint32_t slot, VCell e) {
    if ((e.key & 3u) != ST_SUSPECT) return;
    const uint32_t since1 = v_since1(s, vidx(s, li, slot), e, t);
    if (since1 - 1 + s.S <= t) examine_with(slot, (e.key & ~3u) | ST_DEAD, 1u, false, 0u, HAVE_CELL, e, 0u, 0u);
    else if ((since1 - 1 + s.S) % s.S == row_now) tput(slot + 1);
  };
  SECT(1);                                          // own line
  // phase 1: suspicion deadlines, evaluated on the start-of-tick view
  {
    // A member that just came back up (its cells may be stale, their chains gone) or a cell that says "look
    // everywhere": every view row is a candidate.  The member that came back also rebuilds its cells from what it
    // finds: deadlines still ahead go to the cell of their tick.  The WAVE walks the rows for it, 64 at a time (one
    // lane alone read them one after the other: thousands of dependent round trips per member that woke up, the
    // whole launch waiting for the slowest wave -- 18 ms per tick with 25 rejoins per tick among 2 M members and
    // 6 700 rows in use, profiles/r02o_*): each lane loads one row's cell of the member in turn, the Suspect cells
    // found are handed to the member's own lane, which applies the rule.
    const bool scan_me = act && (woke || (uint32_t)(due.w >> 16) == TR_FULL);
    unsigned long long scanners = __ballot(scan_me);
    if (scanners) {
      PSITE(20);
      if (scan_me) { PSTAT(7); SECT_COUNT(16); }
      if (scan_me && woke) for (uint32_t row = 0; row < s.S; ++row) s.trow[(size_t)row * s.N + li] = make_uint4(0u, 0u, 0u, 0u);
      const uint32_t ns = min(s.g[G_NSLOTS], s.R_phys), lane = tid & 63u;
      for (; scanners; scanners &= scanners - 1ull) {
        const int L = __ffsll((unsigned long long)scanners) - 1;
        const uint32_t li_L = (uint32_t)__builtin_amdgcn_readlane((int)li, L);
        for (uint32_t r0 = 0; r0 < ns; r0 += 64u) {
          const uint32_t r = r0 + lane;
          uint2 e = make_uint2(0u, 0u);
          if (r < ns) e = v_full(s, vidx(s, li_L, r));
          unsigned long long hits = __ballot(r < ns && (e.x & 3u) == ST_SUSPECT && s.slot_used[r < ns ? r : 0u]);
          for (; hits; hits &= hits - 1ull) {
            const int src = __ffsll((unsigned long long)hits) - 1;
            const uint32_t rh = r0 + (uint32_t)src;
            const uint2 eh = make_uint2((uint32_t)__builtin_amdgcn_readlane((int)e.x, src), (uint32_t)__builtin_amdgcn_readlane((int)e.y, src));
            if ((int)lane != L) continue;            // the member's own lane rules on what the wave found
            const uint32_t dl = eh.y - 1 + s.S;
            if (dl <= t) { if (woke || dl == t) examine_with(rh, (eh.x & ~3u) | ST_DEAD, 1u, false, 0u, HAVE_CELL, v_cell_of(eh), 0u, 0u); }
            else if (woke && dl % s.S == row_now) tput(rh + 1);     // a pulled Suspect (since = t): this tick's own cell
            else if (woke) {
              const size_t ix = (size_t)(dl % s.S) * s.N + li;
              const uint4 cell = s.trow[ix];
              TimerCell c2; c2.lo = cell.x | ((unsigned long long)cell.y << 32); c2.hi = cell.z | ((unsigned long long)cell.w << 32); c2.n = 0;
              while (c2.n < TR_PAY && tc_get(cell, c2.n)) c2.n++;
              tc_put_simple(c2, rh + 1);
              s.trow[ix] = tc_pack(c2);
            }
          }
        }
      }
    }
  }
  if (act) {
    if (woke || (uint32_t)(due.w >> 16) == TR_FULL) {
      // (walked by the wave above)
    } else if (timer_due) {
      // the first DB view cells of this tick's cell were loaded above; a slot that is twice in the cell (two
      // suspicions of one subject accepted in one tick) is read again the second time: this thread may just have
      // changed it.  (Loading whole cells and whole explicit-record lines in batches was measured and dropped:
      // 155 us against 141 us without loss, 2.23 against 2.28 ms at 1 % loss, profiles/r02y_*.txt.)
      uint4 cell = due;
      for (bool first = true;; first = false) {
        for (uint32_t k = 0; k < TR_PAY; ++k) {
          const uint32_t v = tc_get(cell, k);
          if (!v) break;
          bool have = first && k < (uint32_t)DB;
          VCell e = dcell[0];
#pragma unroll
          for (int j = 1; j < DB; ++j) if (k == (uint32_t)j) e = dcell[j];
#pragma unroll
          for (int j = 0; j < DB - 1; ++j) have &= !((uint32_t)j < k && dsl[j] == v);
          if (!have && !ABL(ABL_V_LOAD)) e = v_hot(s, vidx(s, li, v - 1));
          if (!ABL(ABL_DEADLINES)) deadline(v - 1, e);
        }
        if (!tc_linked(cell.w)) break;
        SECT_COUNT(17);
        cell = s.tovf[((size_t)row_now * 2u + (par_now ^ 1u)) * s.tovf_cap + tc_link_idx(cell.w)];
      }
    }
    SECT(10);                                       // deadlines
    // phase 2: own probes that ended without any ack: Suspect at the viewed incarnation
    PSITE(21);
    for (uint32_t f = 0; f < nfail; ++f) {
      PSTAT(9);
      const uint32_t j = s.fail[(size_t)li * s.P + f];
      const uint32_t sl = (s.minfo[j] & MI_SLOT) - 1;
      const uint32_t ek = v_key(s, vidx(s, li, sl));
      const uint32_t curk = ek ? ek : s.slot_base[sl];
      const uint32_t key = (curk & ~3u) | ST_SUSPECT;
      if (key > curk) examine(sl, key, 0u, false, 0u);
    }
  }
  // phase 3: rumours received this tick (any order: the merge is commutative).
  SECT(2);                                          // deadlines, failed probes
#if SWIM_MERGE_UNION
  // DECIDE, then BOOK (round 6).  The lanes of a wave are 64 consecutive members and a view row keeps consecutive members next to
  // each other (V[tile][row][256]), so the wave walks the UNION of its lanes' new ring positions in one wave-uniform order: per
  // position ONE predicated load of that row for the lanes that hold the bit -- a run of <= 512 contiguous bytes instead of 64
  // lanes looking at 64 different rows --, the state rule's comparison (src/Core.hs:151-152), the predicated store (saveMember
  // m', :169-179) and one word parked in the lane's LDS column {position, changed-before, key step}.  Nothing else rides in that
  // loop (round 2's union walk dragged the whole bookkeeping through every iteration and lost: 237 us against 188): digest,
  // deadline, queue entry and event of what was ACCEPTED are booked by a per-lane walk over the parked words -- as many steps as
  // the busiest lane accepted, all from LDS (the ring's copy says what a position stands for), no memory round trip.
  {
    constexpr int UB = SWIM_UNION_BATCH;
    unsigned long long fresh = act ? (pushed | pulled) & ~kn : 0ull;
    kn |= fresh;
    if (ABL(ABL_RUMOURS)) fresh = 0;
    unsigned long long U = wave_or64(fresh);          // wave-uniform: the positions anybody of the wave has news at
    uint32_t nacc = 0;
    auto book = [&]() {
      for (uint32_t k = 0; k < nacc; ++k) {
        PSTAT(11);
        const uint32_t w = acc_sh[k][tid], p = w & 63u;
        const uint4 r = ring_sh[p];                  // {slot, key, base, subject} of the id at position p
        account(r.x, r.y, 2u, true, rid_at(p, H) & RID_MASK, w >> 7, ((w >> 6) & 1u) != 0u, r.w, false);
      }
      nacc = 0;
    };
    while (U) {
      PSTAT(10);
      uint32_t p[UB]; uint4 r[UB]; VCell e[UB]; bool has[UB];
      uint32_t n = 0;
#pragma unroll
      for (int k = 0; k < UB; ++k) {
        p[k] = 0; r[k] = make_uint4(0u, 0u, 0u, 0u); e[k] = VCell{0u, 0u}; has[k] = false;
        if (U) {
          p[k] = (uint32_t)__ffsll((unsigned long long)U) - 1u;
          U &= U - 1ull;
          r[k] = ring_sh[p[k]];
          has[k] = ((fresh >> p[k]) & 1ull) != 0ull;
          n = (uint32_t)k + 1u;
        }
      }
#pragma unroll
      for (int k = 0; k < UB; ++k) {
        if (!has[k]) continue;
        if (r[k].x + 1 == my_slot1) {
          // about self -> refute (src/Core.hs:155-166): remember the largest non-Alive incarnation
          if ((r[k].y & 3u) != ST_ALIVE) refute = (refute == NONE32 || (r[k].y >> 2) > refute) ? (r[k].y >> 2) : refute;
          has[k] = false;
        } else if (!ABL(ABL_V_LOAD)) e[k] = v_hot(s, vidx(s, li, r[k].x));
      }
#pragma unroll
      for (int k = 0; k < UB; ++k) {
        if ((uint32_t)k >= n) continue;
        // two rumours about one subject in a batch (Suspect and Dead arriving together; wave-uniform: the rows are): the later
        // one looks at the cell again (this thread's own store is visible to it)
        bool again = false;
#pragma unroll
        for (int j = 0; j < UB; ++j) again |= (j < k) && (r[j].x == r[k].x);
        if (!has[k]) continue;
        PSITE(22 + (k & 1));
        const size_t ix = vidx(s, li, r[k].x);
        if (again && !ABL(ABL_V_LOAD)) e[k] = v_hot(s, ix);
        const uint32_t key = r[k].y, curk = e[k].key ? e[k].key : r[k].z;      // untouched cell: the settled base
        if (key <= curk) continue;                    // old incarnation / weaker state: ignore (:151)
        if (s.strict && (((key & 3u) == ST_SUSPECT && (curk & 3u) != ST_ALIVE) || ((key & 3u) == ST_DEAD && (curk & 3u) == ST_DEAD))) continue;
        PSTAT(5); PSTAT(6); PSTAT(psite); SECT_COUNT(21);
        const bool again_ = v_changed_in(s, ix, e[k], t);               // (before the store below)
        if (!ABL(ABL_V_STORE)) v_put(s, ix, key, t + 1);                // memberLastChange = now (:176)
        acc_sh[nacc][tid] = p[k] | ((uint32_t)again_ << 6) | ((key - curk) << 7);    // (keys are < 2^24: the step fits)
        nacc++;
      }
      if (!U || __ballot(nacc + (uint32_t)UB > (uint32_t)ACC_CAP)) book();   // wave-uniform: a lane's column could overflow in the next batch
    }
  }
#else
  // Each lane walks its own new
  // positions.  (Measured on MI355X, profiles/r02c_variants.txt: a wave-uniform walk over the UNION of the
  // lanes' positions coalesces the view rows but triples the iterations -- 237 us against 188 us; one
  // candidate loop shared by all sources with the bookkeeping parked in LDS -- 258 us.)
  // The loads are batched: a member's next GB positions are decoded together -- what each position stands for
  // comes from the block's copy of the tick's ring in LDS, so the only round trip is the batch's view cells
  // (the kernel waits on such chains most of its time, profiles/r02a_pmc_summary.txt).
  if (act) {
    constexpr int GB = SWIM_GOSSIP_BATCH;
    unsigned long long fresh = (pushed | pulled) & ~kn;
    kn |= fresh;
    if (ABL(ABL_RUMOURS)) fresh = 0;
    for (bool first = true; fresh; first = false) {
      PSTAT(10);
      uint32_t rid[GB]; uint4 r[GB]; VCell e[GB]; uint32_t pos[GB];
      uint32_t n = 0;
#pragma unroll
      for (int k = 0; k < GB; ++k) {
        rid[k] = 0; r[k] = make_uint4(0u, 0u, 0u, 0u); e[k] = VCell{0u, 0u}; pos[k] = NONE32 - 1u;
        if (fresh) {
          const uint32_t p = (uint32_t)__ffsll((unsigned long long)fresh) - 1u;
          fresh &= fresh - 1ull;
          pos[k] = p;
          rid[k] = rid_at(p, H) & RID_MASK;
          r[k] = ring_sh[p];                       // {slot, key, base, subject} of the id at position p
          n = (uint32_t)k + 1u;
        }
      }
#pragma unroll
      for (int k = 0; k < GB; ++k) {
        if ((uint32_t)k >= n || r[k].x + 1 == my_slot1 || ABL(ABL_V_LOAD)) continue;
        // the first batch's cells were asked for a round trip ago (above), unless this thread has stored to one since
        bool got = false;
#pragma unroll
        for (int j = 0; j < GBP; ++j)
          if (first && pf_pos[j] == pos[k] && ((pf_ok >> j) & 1u)) { e[k] = pf_e[j]; got = true; }
        if (!got) e[k] = v_hot(s, vidx(s, li, r[k].x));
      }
#pragma unroll
      for (int k = 0; k < GB; ++k) {
        if ((uint32_t)k >= n) continue;
        PSTAT(11); PSITE(22 + k);
        // two rumours about one subject in a batch (Suspect and Dead arriving together): the later one looks at
        // the cell again (this thread's own store is visible to it)
        bool again = false;
#pragma unroll
        for (int j = 0; j < GB; ++j) again |= (j < k) && (r[j].x == r[k].x);
        examine_with(r[k].x, r[k].y, 2u, true, rid[k], again ? (HAVE_BASE | HAVE_SUBJ) : EX_ALL, e[k], r[k].z, r[k].w);
      }
    }
  }
#endif
  SECT(3);                                          // delivered rumours
  if (act) {
    if (cnt) {
      // explicit records (queues the masks could not carry in full): the records phase has read the sources' lines, filtered
      // their entries through the rings and left the survivors as this member's todo list {slot | rid << 16, key} -- TB
      // entries per round of loads, then their view cells, row bases and subjects in ONE round, then the rule on each
      PSITE(30);
      kn |= s.kn_rec[li];                          // ring positions of the ids the records carried
      const uint4 sg = s.todo_seg[li];             // the list in two segments (records_kernel; one from the phase above)
      auto td = [&](uint32_t x) -> uint2 { return s.todo[x < sg.y ? (size_t)sg.x + x : (size_t)sg.z + (x - sg.y)]; };
      if (s.strict) {
        // strict reference rules: the literal rule is not commutative -- the delivered rumours are applied in the canonical order, by
        // (subject, key) ascending (rows stand for subjects: by (row, key); the order ACROSS subjects changes nothing).  A selection
        // sort over the list where it lies -- the smallest entry above the last one applied, again and again: duplicates cost one
        // pass together; the mode is for checking runs against the reference's rule, not for speed.
        unsigned long long last = 0;
        for (;;) {
          unsigned long long best = ~0ull; uint32_t brid = 0;
          for (uint32_t x = 0; x < cnt; ++x) {
            const uint2 e2 = td(x);
            const unsigned long long k = (((unsigned long long)pe_slot(e2.x) << 32) | e2.y) + 1ull;
            if (k > last && k < best) { best = k; brid = pe_rid(e2.x); }
          }
          if (best == ~0ull) break;
          last = best;
          examine((uint32_t)((best - 1ull) >> 32), (uint32_t)(best - 1ull), 2u, true, brid);
        }
      } else {
      constexpr int TB = SWIM_TODO_BATCH;
      uint2 en[TB];
#pragma unroll
      for (int k = 0; k < TB; ++k) { en[k] = make_uint2(0u, 0u); if ((uint32_t)k < cnt) en[k] = td((uint32_t)k); }
      for (uint32_t x0 = 0; x0 < cnt; x0 += TB) {
        VCell ce[TB]; uint32_t cb[TB], cs[TB];
        uint2 cur_en[TB];
#pragma unroll
        for (int k = 0; k < TB; ++k) {
          cur_en[k] = en[k];
          ce[k] = VCell{0u, 0u}; cb[k] = 0u; cs[k] = 0u;
          const uint32_t slot = pe_slot(en[k].x);
          if (x0 + k < cnt && slot + 1 != my_slot1) {
            if (!ABL(ABL_V_LOAD)) ce[k] = v_hot(s, vidx(s, li, slot));
            cb[k] = s.slot_base[slot]; cs[k] = s.subject_of[slot];
          }
        }
#pragma unroll
        for (int k = 0; k < TB; ++k) { en[k] = make_uint2(0u, 0u); if (x0 + TB + k < cnt) en[k] = td(x0 + TB + k); }   // the next batch's entries travel meanwhile
        for (uint32_t k = 0; k < (uint32_t)TB && x0 + k < cnt; ++k) {
          uint2 e2 = cur_en[0]; VCell e = ce[0]; uint32_t sb = cb[0], sj = cs[0];
          bool again = false;                      // an earlier entry of the batch is about the same subject: look again
#pragma unroll
          for (int j = 1; j < TB; ++j) if (k == (uint32_t)j) { e2 = cur_en[j]; e = ce[j]; sb = cb[j]; sj = cs[j]; }
#pragma unroll
          for (int j = 0; j < TB - 1; ++j) again |= (uint32_t)j < k && pe_slot(cur_en[j].x) == pe_slot(e2.x);
          PSTAT(14);
          examine_with(pe_slot(e2.x), e2.y, 2u, true, pe_rid(e2.x), again ? (HAVE_BASE | HAVE_SUBJ) : EX_ALL, e, sb, sj);
        }
      }
      }
    }
    // ---- refutation: bump own incarnation past the rumour's (src/Core.hs:155-166; D10); rumours at
    // an incarnation below my own are stale and ignored (:151)
    if (refute != NONE32 && refute >= self_inc) {
      uint32_t ni = refute + 1;
      if (ni > INC_MAX) { atomicOr(&s.g[G_ERR], (uint32_t)ERRF_INC); ni = INC_MAX; }
      self_inc = ni;
      PSTAT(15);
      evd += h4(TAG_INC, ((uint64_t)t << 32) | i, ni, 0);
      refutes = 1;
      const uint32_t akey = (ni << 2) | ST_ALIVE;
      if (s.G) s.slot_last[my_slot1 - 1] = t;
      kill_slot(my_slot1 - 1);
      group_put(my_slot1 - 1, find_rid(s, my_slot1 - 1, akey), akey, i);   // Just Alive{..} (:163)
      if (s.event_mask & (1u << 3)) {
        const uint32_t pos = atomicAdd(&s.g[G_EVCUR], 1u);
        if (pos < s.event_cap) s.events[pos] = make_uint4(t, i, i, (akey << 8) | 3u /*REFUTE*/);
        else evdropped++;
      }
    }
    // ---- rebuild the queue: [this tick's group, by subject][aged survivors, order kept], best 8 (D5),
    // assembled in LDS columns, then written as one 64-B line together with its mask.  Every period costs
    // a rumour at least one transmission (age >= 1), so old entries never tie with this tick's group.
    SECT(4);                                        // explicit records, refutation
    const uint32_t age = nsent ? nsent : 1u;
    uint32_t nout = gn;
    unsigned long long qmask = 0;
    uint32_t oow = 0;
    auto publish = [&](uint32_t lo) -> uint32_t {  // mask bit or "cannot express", id parked if too old
      if (rids_off) { oow = MI_OOW; return pe_lo(pe_slot(lo), RID_PARKED); }
      const uint32_t rid = pe_rid(lo);
      if (rid_maskable(rid, H)) qmask |= rid_bit(rid);
      else oow = MI_OOW;
      return park_rid(lo, H);
    };
    for (uint32_t k = 0; k < gn; ++k) {
      asm_[2 * k][tid] = publish(asm_[2 * k][tid]);
      asm_[2 * k + 1][tid] = pe_hi(asm_[2 * k + 1][tid], s.L);
    }
    if (pcount) {
#pragma unroll
      for (int h = 0; h < PB_SLOTS / 2; ++h) {
        const uint4 v = ol[h];                     // (held since the second round of loads; it was read again here)
#pragma unroll
        for (int w = 0; w < 2; ++w) {
          const uint32_t lo = w ? v.z : v.x, hi = w ? v.w : v.y, tx = pe_tx(hi);
          if (tx > age && !((killmask >> (2 * h + w)) & 1u) && nout < (uint32_t)PB_SLOTS) {
            PSTAT(16);
            asm_[2 * nout][tid] = publish(lo);
            asm_[2 * nout + 1][tid] = pe_hi(pe_key(hi), tx - age);
            nout++;
          }
        }
      }
    }
    for (uint32_t k = nout; k < (uint32_t)PB_SLOTS; ++k) { asm_[2 * k][tid] = 0u; asm_[2 * k + 1][tid] = 0u; }
    PSTAT(17, gn);
    if (nout) {
      PSTAT(18);
      wflag = 1u | ((cur ^ 1u) << 1);              // the line itself is stored below, a whole wave at a time
      set_mi(s, i, (mi & ~MI_PB) | (nout << MI_PBN_SHIFT) | ((cur ^ 1u) << 20) | oow);
    } else if (pcount) {
      set_mi(s, i, mi & ~MI_PB);
    }
    SECT(5);                                        // queue rebuilt
    if (!ABL(ABL_STATE_STORES)) {
    { const unsigned long long qm_ = nout ? qmask : 0ull; st_u32x4(reinterpret_cast<uint4*>(&s.pk[li]), make_uint4((uint32_t)qm_, (uint32_t)(qm_ >> 32), (uint32_t)kn, (uint32_t)(kn >> 32))); }
    if (pushed) s.inmask[li] = 0;
    if (timer_due || tnew.n || woke) st_u32x4(&trow_now[li], tc_pack(tnew));   // consumed and refilled in one store
    }
    if (self_inc != hot0.x || woke) s.hot[li] = make_uint2(self_inc, hot0.y & ~1u);
    if (cnt) s.inbox_cnt[li] = 0;
    pb_writes = (pcount || nout) ? 1u : 0u;
  }
  ctr_add_wave(&sh, C_CHANGES, changes);
  ctr_add_wave(&sh, C_PB_WRITES, pb_writes);
  {
    const unsigned long long wevd = wave_sum64(evd);
    if ((tid & 63u) == 0u && wevd) atomicAdd(&sh.evd, wevd);
  }
  if (__ballot((timers_fired | refutes | evdropped) != 0u)) {   // the rare ones: a wave-uniform test first
    ctr_add_wave(&sh, C_TIMERS_FIRED, timers_fired & 0xFFFFu);   // a lane fires at most one timer per row: < 65 535
    ctr_add_wave(&sh, C_FALSE_DEADS, timers_fired >> 16);
    ctr_add_wave(&sh, C_REFUTES, refutes);
    ctr_add_wave(&sh, C_EVENTS_DROPPED, evdropped);
  }
  // ---- store the rebuilt lines.  L2 does not merge a lane's four 16-B pieces into one fabric write, so
  // the wave stores its 64 lines transposed: instruction k, lane l writes piece (l & 3) of the line of
  // member 16 k + (l >> 2): every instruction covers 1 KB of contiguous memory in full 128-B lines.
  SECT(6);                                          // state stores, counters
  wfl[tid] = wflag;
  lds_wave_sync();                                  // a wave stores its own 64 lines: nothing crosses waves here
  SECT(7);                                          // barrier
  {
    const uint32_t wbase = tid & ~63u, lane = tid & 63u, q = lane & 3u;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t m = wbase + 16u * k + (lane >> 2);          // thread whose line this lane helps to store
      const uint32_t fl = wfl[m];
      if ((fl & 1u) && !ABL(ABL_LINE_STORE)) {
#if SWIM_MERGE_SORT
        const uint32_t gm = blockIdx.x * BLOCK + perm_sh[m];         // the member thread m stepped
#else
        const uint32_t gm = blockIdx.x * BLOCK + m;
#endif
        uint4* line = reinterpret_cast<uint4*>(s.pb + ((size_t)(fl >> 1) * s.N + gm) * PB_SLOTS);
        st_u32x4(&line[q], make_uint4(asm_[4 * q][m], asm_[4 * q + 1][m], asm_[4 * q + 2][m], asm_[4 * q + 3][m]));
      }
    }
  }
  SECT(8);                                          // line store
  ctr_flush(s, &sh, blockIdx.x);
  SECT(9);
}

// ================================================================================================
// cross-shard exchange kernels (n_shards > 1)
// ================================================================================================
// One tick of a shard:  begin_kernel (faults, window head, ring, ring dictionary) -> publish_kernel (my slice of the
// replicas) -> ROUND 1: all-gather of dictionary + lists (r), queue masks, queue bytes -> xlat_kernel -> probe_kernel<.., true>
// (routes its own records) -> ROUND 2: all-to-all-v of {dst, src} records (q) -> ingest_kernel -> merge_kernel.

// my slice of the replicas, as the peers' probes of this tick will read it: every member's queue mask (over MY ring of the
// tick: the dictionary begin_kernel wrote) and queue byte; a queue that cannot travel as a mask -- an entry outside the mask
// window, or a tick in which my masks are off (a burst of rumour ids: swim_device.h) -- is published as a LIST of (subject,
// key) behind the dictionary, the same list for every peer.  One thread per member, one list reservation per wave.
__global__ __launch_bounds__(BLOCK) void publish_kernel(DevState s, uint32_t t) {
  const uint32_t Hprev = s.g[G_PREV], H = s.g[G_HEAD];
  const bool use_mask = H - Hprev <= MASK_SLACK && !s.strict;   // (strict reference rules: no delivery is ever filtered -- every queue travels as a list)
  for (uint32_t l0 = blockIdx.x * BLOCK; l0 < s.N; l0 += gridDim.x * BLOCK) {      // (wave-uniform trip count)
    const uint32_t li = l0 + threadIdx.x, i = s.lo + li;
    uint32_t mi = 0;
    bool list = false;
    if (li < s.N) {
      mi = s.minfo[i];
      const uint32_t n = mi_up(mi) ? mi_pbn(mi) : 0u;
      list = n != 0u && (!use_mask || (mi & MI_OOW));
      s.mask_all[i] = (n && !list) ? s.pk[li].x : 0ull;
      s.q_all[i] = (uint8_t)(n | (list ? Q_OOW : 0u));
    }
    const unsigned long long b = __ballot(list);
    if (!b) continue;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t base = 0;
    if (lane == 0u) base = atomicAdd(&s.g[G_XLINES], (uint32_t)__popcll(b));
    base = (uint32_t)__builtin_amdgcn_readlane((int)base, 0);
    if (!list) continue;
    const uint32_t pos = base + (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
    if ((pos + 1u) * XLINE_RECS > s.r_cap) { atomicOr(&s.g[G_ERR], (uint32_t)ERRF_XCHG); continue; }
    uint4* rec = s.r_send + DICT_RECS + (size_t)pos * XLINE_RECS;
    const uint2* line = reinterpret_cast<const uint2*>(line_ptr(s, mi_buf(mi), li));
    uint32_t w[2 * PB_SLOTS];
    uint32_t n = 0;
#pragma unroll
    for (int k = 0; k < PB_SLOTS; ++k) {
      const uint2 e = line[k];
      const bool v = pe_tx(e.y) != 0u;
      w[2 * k] = v ? s.subject_of[pe_slot(e.x)] : 0u;
      w[2 * k + 1] = v ? pe_key(e.y) : 0u;
      n += v ? 1u : 0u;                              // (a line holds its valid entries first)
    }
    rec[0] = make_uint4(i, n, t, 0u);
#pragma unroll
    for (int k = 0; k < PB_SLOTS / 2; ++k) rec[1 + k] = make_uint4(w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]);
  }
}

// the round-1 segment of peer `peer` (dictionary, then lists) and its number of lists, wherever they lie (PeerView)
__device__ inline const uint4* r_list_of(const DevState& s, const PeerView& pv, const PeerCounts& rc, uint32_t peer, uint32_t* nlists) {
  if (pv.direct) { *nlists = min(*pv.rn[peer], s.r_cap / XLINE_RECS); return pv.r[peer]; }
  const uint32_t got = rc.v[peer];
  *nlists = got > DICT_RECS ? min(got - DICT_RECS, s.r_cap) / XLINE_RECS : 0u;
  return got >= DICT_RECS ? s.r_recv + (size_t)peer * (DICT_RECS + s.r_cap) : nullptr;
}

// before the probes, one launch of 256-thread blocks:
//   blocks [0, n_shards): every peer's ring dictionary of the tick in MY numbering (slots and rumour ids are per shard).
//     Rumours this shard never heard of get their slot and id here; such ids are younger than this tick's head: my masks
//     cannot carry them until the next tick (position 0xFF: the delivery goes through a foreign line);
//   blocks [n_shards, n_shards + XLAT_INDEX_BLOCKS): where each received list lies (xidx), for the lookups of ingest_kernel;
//   the rest (direct mode only): the peers' slices of the replicas -- queue masks and queue bytes -- copied over.
constexpr uint32_t XLAT_INDEX_BLOCKS = 32;
__global__ __launch_bounds__(BLOCK) void xlat_kernel(DevState s, uint32_t t, PeerCounts r_counts, PeerView pv) {
  const uint32_t H = s.g[G_HEAD];
  const bool use_mask = H - s.g[G_PREV] <= MASK_SLACK && !s.strict;
  if (blockIdx.x >= s.n_shards + XLAT_INDEX_BLOCKS) {
    // the replicas: everybody else's slice from its owner (8 + 1 bytes per member, coalesced)
    const uint32_t nb = gridDim.x - s.n_shards - XLAT_INDEX_BLOCKS, b = blockIdx.x - s.n_shards - XLAT_INDEX_BLOCKS;
    for (uint32_t peer = 0; peer < s.n_shards; ++peer) {
      if (peer == s.shard) continue;
      const unsigned long long* m = pv.mask[peer]; const uint8_t* q = pv.qb[peer];
      const uint32_t lo = peer * s.N;
      for (uint32_t k = b * BLOCK + threadIdx.x; k < s.N; k += nb * BLOCK) s.mask_all[lo + k] = m[lo + k];
      const uint32_t* q4 = reinterpret_cast<const uint32_t*>(q + lo);       // (N is a multiple of 4 here, or the tail goes bytewise)
      uint32_t* d4 = reinterpret_cast<uint32_t*>(s.q_all + lo);
      if ((lo & 3u) == 0u) {
        for (uint32_t k = b * BLOCK + threadIdx.x; k < s.N / 4u; k += nb * BLOCK) d4[k] = q4[k];
        for (uint32_t k = (s.N & ~3u) + b * BLOCK + threadIdx.x; k < s.N; k += nb * BLOCK) s.q_all[lo + k] = q[lo + k];
      } else {
        for (uint32_t k = b * BLOCK + threadIdx.x; k < s.N; k += nb * BLOCK) s.q_all[lo + k] = q[lo + k];
      }
    }
    return;
  }
  if (blockIdx.x >= s.n_shards) {
    for (uint32_t peer = 0; peer < s.n_shards; ++peer) {
      if (peer == s.shard) continue;
      uint32_t nl = 0;
      const uint4* seg = r_list_of(s, pv, r_counts, peer, &nl);
      for (uint32_t k = (blockIdx.x - s.n_shards) * BLOCK + threadIdx.x; k < nl; k += XLAT_INDEX_BLOCKS * BLOCK) {
        const uint4 hd = seg[DICT_RECS + (size_t)k * XLINE_RECS];
        if (hd.x < s.NT && hd.z == t) s.xidx[hd.x] = k;
      }
    }
    return;
  }
  const uint32_t peer = blockIdx.x, p = threadIdx.x;
  if (p >= DICT_ENTRIES) return;
  uint2 out = make_uint2(NONE32, 0u);
  uint32_t nl = 0;
  const uint4* seg = peer != s.shard ? r_list_of(s, pv, r_counts, peer, &nl) : nullptr;
  if (seg) {
    const uint2 e = reinterpret_cast<const uint2*>(seg)[p];   // {subject, key}
    if (e.x != NONE32) {
      const uint32_t slot = get_slot(s, e.x);
      // a dictionary changes by a few entries per tick: keep last tick's translation of this position while it
      // still names the same rumour (find_rid would take a fresh id on every call for a rumour whose cache way
      // a newer rumour about the subject has taken over -- an id burst per tick that switches the masks off)
      const uint2 prev = s.xl[(size_t)peer * DICT_ENTRIES + p];
      const uint32_t prid = pe_rid(prev.x);
      const uint2 pr = s.rum[prid & RID_MASK];
      uint2 o;
      if (prev.x != NONE32 && pe_key(prev.y) == e.y && pe_slot(prev.x) == slot && prid != RID_PARKED && pr.x == slot && pr.y == e.y &&
          ((s.g[G_NRUM] - prid) & RID_MASK) < RID_FAR)
        o = make_uint2(prev.x, e.y);
      else {
        uint32_t num = 0;
        const uint32_t nrid = find_rid(s, slot, e.y, &num);
        o = make_uint2(pe_lo(slot, young_rid(nrid, num, H)), e.y);
      }
      const uint32_t rid = pe_rid(o.x);
      out = make_uint2(o.x, o.y | ((use_mask && rid_in_ring(rid, H)) ? ((rid & 63u) << 24) : (0xFFu << 24)));
    }
  }
  s.xl[(size_t)peer * DICT_ENTRIES + p] = out;
}

// a foreign line for local member dst_li: `nf` entries {slot | rid << 16, key | tx = 1 << 24} my masks cannot carry, read
// through an explicit record (the exact path behind the masks: records phase / records_kernel)
__device__ inline void push_foreign(const DevState& s, uint32_t t, uint32_t dst_li, const uint32_t* ent, uint32_t nf) {
  const uint32_t k = atomicAdd(&s.g[G_FLDYN], 1u);
  if (k >= s.fl_dyn_cap) { atomicOr(&s.g[G_ERR], (uint32_t)ERRF_XCHG); return; }
  uint32_t* fl = reinterpret_cast<uint32_t*>(s.fl + ((size_t)s.fl_dyn_base + k) * 4);
  for (uint32_t e = 0; e < (uint32_t)PB_SLOTS; ++e) { fl[2 * e] = e < nf ? ent[2 * e] : 0u; fl[2 * e + 1] = e < nf ? ent[2 * e + 1] : 0u; }
  __threadfence();
  push(s, t, dst_li, SRC_FOREIGN | (s.fl_dyn_base + k));
  s.g[G_ANYREC] = t + 1u;                            // same value from every writer: the records phase has work
}

// after round 2: the records {dst, src} for my members -- "dst merges src's start-of-tick queue" -- from every shard's
// probes (mine included: segment [shard] of q_send never left).  src local: its queue mask as any local delivery.  src
// remote: its replicated mask through its owner's dictionary; entries my masks cannot carry this tick, and queues that
// travel as lists, become foreign lines.
__global__ __launch_bounds__(BLOCK) void ingest_kernel(DevState s, uint32_t t, PeerCounts q_counts, PeerView pv) {
  __shared__ uint2 xls[MAX_SHARDS * DICT_ENTRIES];
  __shared__ uint32_t pref[MAX_SHARDS + 1];           // the peers' segments as ONE index space (a pass per peer made the kernel
  for (uint32_t k = threadIdx.x; k < s.n_shards * DICT_ENTRIES; k += BLOCK) xls[k] = s.xl[k];   // wait out its round trips G times)
  if (threadIdx.x == 0) {
    uint32_t acc = 0;
    for (uint32_t peer = 0; peer < s.n_shards; ++peer) {
      pref[peer] = acc;
      acc += min(peer == s.shard ? s.send_cnt[MAX_SHARDS + peer] : (pv.direct ? *pv.qn[peer] : q_counts.v[peer]), s.p_cap);
    }
    pref[s.n_shards] = acc;
  }
  __syncthreads();
  const uint32_t Hprev = s.g[G_PREV], H = s.g[G_HEAD];
  const bool use_mask = H - Hprev <= MASK_SLACK && !s.strict;
  const unsigned long long stale = stale_positions(Hprev, H);
  const uint32_t total = pref[s.n_shards];
  auto record_at = [&](uint32_t k) -> uint2 {
    uint32_t peer = 0;
    while (peer + 1u < s.n_shards && k >= pref[peer + 1u]) ++peer;
    const uint2* list = peer == s.shard ? s.q_send + (size_t)peer * s.p_cap : (pv.direct ? pv.q[peer] : s.q_recv + (size_t)peer * s.p_cap);
    return list[k - pref[peer]];
  };
  // the rare forms of a delivery, one record at a time: a local source (a peer's prober walked a chain through two of my
  // members), a queue that travels as a list, a mask with entries my masks cannot carry
  auto slow = [&](uint32_t dl, uint32_t src, uint32_t qs) {
    if (is_local(s, src)) {
      const uint32_t ms = s.minfo[src];
      if (!mi_pbn(ms)) return;
      deliver_local(s, t, use_mask, stale, dl, src - s.lo, ms, use_mask ? s.pk[src - s.lo].x : 0ull);
      if (!use_mask || (ms & MI_OOW)) s.g[G_ANYREC] = t + 1u;
      return;
    }
    unsigned long long bits = 0;
    uint32_t ent[2 * PB_SLOTS], nf = 0;
    if (!(qs & Q_OOW)) {
      unsigned long long m = s.mask_all[src];
      const uint2* d = xls + owner_of(s, src) * DICT_ENTRIES;
      while (m) {
        const uint32_t q = (uint32_t)__ffsll((unsigned long long)m) - 1u;
        m &= m - 1ull;
        const uint2 e = d[q];
        if (e.x == NONE32) continue;               // cannot happen: the owner set the bit from an entry of its ring
        if ((e.y >> 24) != 0xFFu) bits |= 1ull << (e.y >> 24);
        else if (nf < (uint32_t)PB_SLOTS) { ent[2 * nf] = e.x; ent[2 * nf + 1] = pe_hi(pe_key(e.y), 1u); nf++; }
      }
    } else {
      // the queue as a list of (subject, key): where it lies was noted by xlat_kernel; the list carries its member and tick
      const uint32_t at = s.xidx[src], own = owner_of(s, src);
      const uint4* seg = pv.direct ? pv.r[own] : s.r_recv + (size_t)own * (DICT_RECS + s.r_cap);
      const uint4* rec = seg + DICT_RECS + (size_t)at * XLINE_RECS;
      const uint4 hd = at < s.r_cap / XLINE_RECS ? rec[0] : make_uint4(NONE32, 0u, 0u, 0u);
      if (hd.x != src || hd.z != t) { atomicOr(&s.g[G_ERR], (uint32_t)ERRF_XCHG); return; }
      const uint32_t ne = min(hd.y, (uint32_t)PB_SLOTS);
      const uint2* pe = reinterpret_cast<const uint2*>(rec + 1);
      for (uint32_t e = 0; e < ne; ++e) {
        const uint2 sk = pe[e];
        const uint32_t slot = get_slot(s, sk.x);
        uint32_t num = 0;
        const uint32_t rid0 = find_rid(s, slot, sk.y, &num), rid = young_rid(rid0, num, H);
        if (use_mask && rid_in_ring(rid, H)) bits |= rid_bit(rid);     // an id of an earlier tick
        else { ent[2 * nf] = pe_lo(slot, rid); ent[2 * nf + 1] = pe_hi(sk.y, 1u); nf++; }
      }
    }
    if (bits) {
      const unsigned long long mm = bits & ~(s.pk[dl].y & ~stale);
      if (mm) atomicOr(&s.inmask[dl], mm);
    }
    if (nf) push_foreign(s, t, dl, ent, nf);
  };
  // the common form -- a remote source whose queue is a mask my ring can carry in full -- two records per thread and round:
  // records, then queue bytes, then masks and known-rings, each round of loads issued together
  constexpr int U = 2;
  const uint32_t stride = gridDim.x * BLOCK;
  for (uint32_t k0 = blockIdx.x * BLOCK + threadIdx.x; k0 < total; k0 += stride * U) {
    uint2 o[U]; uint32_t qs[U]; bool live[U], fast[U];
    unsigned long long m[U], kn[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t k = k0 + (uint32_t)u * stride;
      live[u] = k < total;
      o[u] = live[u] ? record_at(k) : make_uint2(NONE32, NONE32);
      live[u] = live[u] && is_local(s, o[u].x) && o[u].y < s.NT;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      qs[u] = (live[u] && !is_local(s, o[u].y)) ? (uint32_t)s.q_all[o[u].y] : 0u;
      fast[u] = live[u] && !is_local(s, o[u].y) && (qs[u] & Q_PBN) && !(qs[u] & Q_OOW) && use_mask;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      m[u] = fast[u] ? s.mask_all[o[u].y] : 0ull;
      kn[u] = fast[u] ? s.pk[o[u].x - s.lo].y : 0ull;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (!live[u]) continue;
      const uint32_t dl = o[u].x - s.lo, src = o[u].y;
      if (fast[u]) {
        const uint2* d = xls + owner_of(s, src) * DICT_ENTRIES;
        unsigned long long bits = 0, mm = m[u];
        bool whole = true;
        while (mm) {
          const uint32_t q = (uint32_t)__ffsll((unsigned long long)mm) - 1u;
          mm &= mm - 1ull;
          const uint2 e = d[q];
          if (e.x == NONE32) continue;
          if ((e.y >> 24) == 0xFFu) { whole = false; break; }
          bits |= 1ull << (e.y >> 24);
        }
        if (whole) {
          const unsigned long long news = bits & ~(kn[u] & ~stale);
          if (news) atomicOr(&s.inmask[dl], news);
          continue;
        }
      }
      if (is_local(s, src) || (qs[u] & Q_PBN)) slow(dl, src, qs[u]);
    }
  }
}

// ================================================================================================
// auxiliary kernels
// ================================================================================================

// Settling, the bookkeeping part (one block; include/swimsim.h).  After merge_kernel of
// tick u: (a) the rows it cleared go to the free stack; (b) every row of its eligible list whose subject
// was quiet during u as well is committed: base := max(base, largest entry among the members that were
// up), the subject loses its row (minfo), the row joins the list the NEXT merge clears.  Nothing is
// settled while an up member holds the subject Suspect.  Runs at the start of begin_kernel, or on its own
// before state is read (digest, views) -- whichever comes first.
__device__ inline void settle_finish(const DevState& s) {
  if (!s.G || s.n_shards > 1 || !s.g[G_SETTLE_PENDING]) return;     // uniform; shards: settle_publish / settle_commit
  __shared__ uint32_t nz_new, nfree;
  const uint32_t u = s.g[G_SETTLE_TICK], ns = s.g[G_SETTLE_N], nz = s.g[G_ZERO_N];
  if (threadIdx.x == 0) { nz_new = 0; nfree = s.g[G_NFREE]; }
  __syncthreads();
  for (uint32_t k = threadIdx.x; k < nz; k += blockDim.x) s.free_rows[nfree + k] = s.zero_slots[k];
  __syncthreads();
  // largest entry among the up members of each eligible row: reduce merge_kernel's per-block partials
  __shared__ uint32_t red[BLOCK];
  for (uint32_t k = 0; k < ns; ++k) {
    uint32_t m = 0;
    for (uint32_t b = threadIdx.x; b < s.nblocks; b += blockDim.x) m = max(m, s.settle_part[(size_t)k * s.nblocks + b]);
    red[threadIdx.x] = m;
    __syncthreads();
    for (uint32_t o = blockDim.x / 2; o > 0; o >>= 1) {
      if (threadIdx.x < o) red[threadIdx.x] = max(red[threadIdx.x], red[threadIdx.x + o]);
      __syncthreads();
    }
    if (threadIdx.x == 0) s.settle_key[k] = red[0];
    __syncthreads();
  }
  unsigned settled = 0;
  for (uint32_t k = threadIdx.x; k < ns; k += blockDim.x) {
    const uint32_t slot = s.settle_slots[k], kmax = s.settle_key[k], last = s.slot_last[slot];
    if (last != NONE32 && u - last < s.G) continue;          // somebody changed its mind during tick u
    if ((kmax & 3u) == ST_SUSPECT) continue;                 // a timer is still running
    const uint32_t subject = s.subject_of[slot];
    const uint32_t nb = max(s.base_key[subject], kmax);
    s.base_key[subject] = nb;
    s.base_since[subject] = u;
    set_mi(s, subject, (s.minfo[subject] & ~(MI_SLOT | MI_BASE)) | ((nb & 3u) << MI_BASE_SHIFT));
    s.slot_used[slot] = 0;
    for (int w = 0; w < RT_WAYS; ++w) s.rtab[(size_t)slot * RT_WAYS + w] = 0ull;
    s.zero_slots[atomicAdd(&nz_new, 1u)] = slot;             // read above by this block only, behind the barrier
    settled++;
  }
  if (settled) {
    atomicAdd(&s.g[G_NLIVE], 0u - settled);
    atomicAdd(reinterpret_cast<unsigned long long*>(&s.blk[(size_t)s.nblocks * C_COUNT + C_SETTLED]), (unsigned long long)settled);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    s.g[G_NFREE] = nfree + nz; s.g[G_ZERO_N] = nz_new; s.g[G_SETTLE_N] = 0; s.g[G_SETTLE_PENDING] = 0;
  }
  __syncthreads();
}
__global__ __launch_bounds__(BLOCK) void settle_flush_kernel(DevState s) { settle_finish(s); }

// Settling on a sharded cluster.  A subject settles when it is quiet on EVERY shard and the
// largest entry among ALL up members is not Suspect, and every shard must commit the same base in the same
// tick (the base is everybody's default).  So after merge_kernel of tick u each shard publishes what its rows
// say -- one 8-byte record per row that is a candidate here (quiet for G ticks, with the largest entry among
// my up members) or a veto (changed / announced within G ticks) --, the lists are all-gathered (exchange round
// 3) and settle_commit_kernel takes the same decision everywhere.  Rows that hold nothing and are neither
// (opened this tick for a peer's dictionary entry) say nothing.
__global__ __launch_bounds__(BLOCK) void settle_publish_kernel(DevState s, uint32_t u) {
  __shared__ uint32_t nrec, nfree;
  __shared__ uint32_t red[BLOCK];
  const uint32_t ns = s.g[G_SETTLE_N], nz = s.g[G_ZERO_N];
  if (threadIdx.x == 0) { nrec = 0; nfree = s.g[G_NFREE]; }
  __syncthreads();
  for (uint32_t k = threadIdx.x; k < nz; k += blockDim.x) s.free_rows[nfree + k] = s.zero_slots[k];   // cleared by this tick's merge
  for (uint32_t k = 0; k < ns; ++k) {
    uint32_t m = 0;
    for (uint32_t b = threadIdx.x; b < s.nblocks; b += blockDim.x) m = max(m, s.settle_part[(size_t)k * s.nblocks + b]);
    red[threadIdx.x] = m;
    __syncthreads();
    for (uint32_t o = blockDim.x / 2; o > 0; o >>= 1) {
      if (threadIdx.x < o) red[threadIdx.x] = max(red[threadIdx.x], red[threadIdx.x + o]);
      __syncthreads();
    }
    if (threadIdx.x == 0) s.settle_key[k] = red[0];
    __syncthreads();
  }
  auto emit = [&](uint32_t x, uint32_t y) {
    const uint32_t pos = atomicAdd(&nrec, 1u);
    if (pos >= s.s_cap) { atomicOr(&s.g[G_ERR], (uint32_t)ERRF_XCHG); return; }
    for (uint32_t g = 0; g < s.n_shards; ++g) s.s_send[(size_t)g * s.s_cap + pos] = make_uint2(x, y);
  };
  for (uint32_t k = threadIdx.x; k < ns; k += blockDim.x) {
    const uint32_t slot = s.settle_slots[k], last = s.slot_last[slot];
    if (last != NONE32 && u - last < s.G) continue;          // somebody changed its mind during tick u: a veto below
    emit(s.subject_of[slot] | SR_CAND, s.settle_key[k] & SR_KEY);
  }
  const uint32_t nrows = min(s.g[G_NSLOTS], s.R_phys);
  for (uint32_t r = threadIdx.x; r < nrows; r += blockDim.x) {
    if (!s.slot_used[r]) continue;
    const uint32_t last = s.slot_last[r];
    if (last != NONE32 && u - last < s.G) emit(s.subject_of[r] | SR_VETO, 0u);
  }
  __syncthreads();
  if (threadIdx.x == 0) { s.g[G_NFREE] = nfree + nz; s.g[G_ZERO_N] = 0; s.g[G_SETTLE_SEND] = min(nrec, s.s_cap); }
}

// after round 3: every shard's list (mine in s_send, the peers' in s_recv).  One u32 per subject collects the
// lists with atomicMax: a veto outranks everything, candidates combine to the largest entry; the thread that
// takes the word back (atomicExch) commits the subject -- once per shard, the same decision everywhere.
__global__ __launch_bounds__(BLOCK) void settle_commit_kernel(DevState s, uint32_t u, PeerCounts counts, PeerView pv) {
  __shared__ uint32_t nz_new;
  if (threadIdx.x == 0) nz_new = 0;
  auto list_of = [&](uint32_t p, uint32_t* n) -> const uint2* {
    if (p == s.shard) { *n = s.g[G_SETTLE_SEND]; return s.s_send + (size_t)p * s.s_cap; }
    if (pv.direct) { *n = min(*pv.stn[p], s.s_cap); return pv.st[p]; }      // the peer's list where it lies (swimsim_cluster_step)
    *n = min(counts.v[p], s.s_cap);
    return s.s_recv + (size_t)p * s.s_cap;
  };
  for (uint32_t p = 0; p < s.n_shards; ++p) {
    uint32_t n; const uint2* l = list_of(p, &n);
    for (uint32_t k = threadIdx.x; k < n; k += blockDim.x) {
      const uint2 r = l[k];
      const uint32_t subject = r.x & ID_MASK;
      if (subject < s.NT) atomicMax(&s.settle_acc[subject], (r.x & (SR_CAND | SR_VETO)) | (r.y & SR_KEY));
    }
  }
  __syncthreads();
  unsigned settled = 0, released = 0;
  for (uint32_t p = 0; p < s.n_shards; ++p) {
    uint32_t n; const uint2* l = list_of(p, &n);
    for (uint32_t k = threadIdx.x; k < n; k += blockDim.x) {
      const uint32_t subject = l[k].x & ID_MASK;
      if (subject >= s.NT) continue;
      const uint32_t v = atomicExch(&s.settle_acc[subject], 0u);
      if (!(v & SR_CAND) || (v & SR_VETO)) continue;          // taken by another thread / not quiet everywhere
      const uint32_t kmax = v & SR_KEY;
      if ((kmax & 3u) == ST_SUSPECT) continue;                 // a timer is still running somewhere
      const uint32_t nb = max(s.base_key[subject], kmax);
      s.base_key[subject] = nb;
      s.base_since[subject] = u;
      const uint32_t mi = s.minfo[subject], row1 = mi & MI_SLOT;
      set_mi(s, subject, (mi & ~(MI_SLOT | MI_BASE)) | ((nb & 3u) << MI_BASE_SHIFT));
      if (row1) {                                              // my row of the subject goes back
        const uint32_t slot = row1 - 1u;
        s.slot_used[slot] = 0;
        for (int w = 0; w < RT_WAYS; ++w) s.rtab[(size_t)slot * RT_WAYS + w] = 0ull;
        s.zero_slots[atomicAdd(&nz_new, 1u)] = slot;
        released++;
      }
      if (is_local(s, subject)) settled++;                     // counted once per cluster: by the subject's owner
    }
  }
  if (released) atomicAdd(&s.g[G_NLIVE], 0u - released);
  if (settled) atomicAdd(reinterpret_cast<unsigned long long*>(&s.blk[(size_t)s.nblocks * C_COUNT + C_SETTLED]), (unsigned long long)settled);
  __syncthreads();
  if (threadIdx.x == 0) { s.g[G_ZERO_N] = nz_new; s.g[G_SETTLE_N] = 0; s.g[G_SETTLE_PENDING] = 0; }
}

// does member c have a change scheduled in the tick whose changes are `faults` (sorted by member)?
__device__ inline bool changes_this_tick(const FaultRec* faults, uint32_t nfaults, uint32_t c) {
  uint32_t lo = 0, hi = nfaults;
  while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (faults[mid].member < c) lo = mid + 1; else hi = mid; }
  return lo < nfaults && faults[lo].member == c;
}

// the host member m pulls from in tick t (include/swimsim.h, "Join-time state pull" / "Periodic state pull"; purpose =
// P_JOIN / P_PULL): the first of 8 draws that is not m, has no change scheduled in this tick, is up -- its state before
// the tick -- and, with periodic pulls on, is not one of the tick's pullers (nobody reads a map that is being written).
// A pure function of replicated data (hashes, the schedule, ground truth): every shard finds the same host.
__device__ inline uint32_t pull_host(const DevState& s, uint32_t t, uint32_t tk, uint32_t m, const FaultRec* faults, uint32_t nfaults,
                                     uint32_t purpose) {
  const uint32_t mk = mix32(tk ^ m);
  for (uint32_t a = 0; a < SEL_ATTEMPTS; ++a) {
    const uint32_t c = __umulhi(hash_mk(mk, (purpose << 24) | a, 0), s.NT);
    if (c == m) continue;
    if (s.pull_T && c % s.pull_T == t % s.pull_T) continue;
    if (changes_this_tick(faults, nfaults, c)) continue;
    if (!mi_up(s.minfo[c])) continue;
    return c;
  }
  return NONE32;
}
__device__ inline uint32_t join_host(const DevState& s, uint32_t t, uint32_t tk, uint32_t m, const FaultRec* faults, uint32_t nfaults) {
  return pull_host(s, t, tk, m, faults, nfaults, P_JOIN);
}

// one pulled entry: joiner ml's cell of `slot` becomes kh if that is news to it
constexpr uint32_t PUSH_REC = 1u << 31;             // round-0 record {member | PUSH_REC, subject, entry}: a PUSH into a host's map (ids are < 2^27 on shards)
// the push half of a push-pull for one entry of a puller's map (km about `subject`): the host's entry is raised to it.  Several pullers
// -- blocks of push_kernel, records from other shards -- may raise one entry: atomicMax on the key, whoever raises it accounts for its
// step (the event digest is linear in the key), the first to stamp lastChange counts the change
__device__ inline void push_entry(const DevState& s, uint32_t t, uint32_t host, uint32_t slot, uint32_t subject, uint32_t km,
                                  unsigned long long* evd, unsigned* pushed, unsigned* suspects) {
  if (!km || km <= s.slot_base[slot]) return;       // (an untouched cell of the host reads 0: never raise it to below the base)
  const size_t ix = vidx(s, host - s.lo, slot);
  const uint32_t old = v_raise_key(s, ix, km, t);
  const uint32_t curk = old ? old : s.slot_base[slot];
  if (km <= curk) return;
  *evd += (mix64(mix64(mix64((uint64_t)TAG_EV) + (((uint64_t)t << 32) | host)) + subject) | 1ull) * (unsigned long long)(km - curk);
  if (v_stamp(s, ix, t) != t + 1u) (*pushed)++;
  if (s.G) s.slot_last[slot] = t;
  *suspects += (km & 3u) == ST_SUSPECT ? 1u : 0u;
}
__device__ inline void pull_entry(const DevState& s, uint32_t t, uint32_t mbr, uint32_t slot, uint32_t subject, uint32_t kh,
                                  unsigned long long* evd, unsigned* pulled) {
  const size_t ix = vidx(s, mbr - s.lo, slot);
  const uint32_t vm = v_key(s, ix), curk = vm ? vm : s.slot_base[slot];
  if (kh <= curk) return;
  v_put(s, ix, kh, t + 1);                             // its deadline, if Suspect: the cells are rebuilt by merge_kernel
  if (s.G) s.slot_last[slot] = t;
  *evd += (mix64(mix64(mix64((uint64_t)TAG_EV) + (((uint64_t)t << 32) | mbr)) + subject) | 1ull) * (unsigned long long)(kh - curk);
  (*pulled)++;
}

// Start of tick t, one block, in two parts (bit 0 / bit 1 of `part`; one launch does both unless a sharded
// cluster has join-time pulls to exchange in between):
//   A  settling bookkeeping of the tick before, then the ground-truth changes scheduled for t (host-sorted by member
//      within the tick: one thread applies all changes of one member in order, members in parallel); the members that
//      came up are listed in `joined` (every shard lists the same ones: ground truth is replicated);
//   B  the joiners' state pulls -- from hosts on this shard directly, from hosts elsewhere as the records their owners
//      sent (pull_send_kernel, exchange round 0) --, then the snapshot of the rumour-id counter that fixes the tick's
//      window head H (no ids are allocated between here and merge_kernel), the tick's ring and dictionary, the rows
//      eligible for settling at the end of t.
__global__ __launch_bounds__(BLOCK) void begin_kernel(DevState s, uint32_t t, uint32_t tk, const FaultRec* faults, uint32_t nfaults,
                                                      uint32_t* joined, uint32_t part, PeerCounts jc, JoinView jv) {
  __shared__ unsigned long long evd_sh;
  __shared__ unsigned dropped_sh, nset, changes_sh;
  if (part & 1u) settle_finish(s);
  if (threadIdx.x == 0) {
    evd_sh = 0; dropped_sh = 0; nset = 0; changes_sh = 0;
    if (part & 1u) { s.g[G_NJOINED] = 0; for (int g = 0; g < MAX_SHARDS; ++g) s.g[G_JSEND + g] = 0; }
  }
  __syncthreads();
  if (part & 1u)
  for (uint32_t k0 = threadIdx.x; k0 < nfaults; k0 += blockDim.x) {
    if (k0 && faults[k0 - 1].member == faults[k0].member) continue;      // not the first change of its member
    unsigned long long evd = 0; unsigned dropped = 0;
    bool came_up = false;
    for (uint32_t k = k0; k < nfaults && faults[k].member == faults[k0].member; ++k) {
      const uint32_t mbr = faults[k].member, up = faults[k].up;
      uint32_t mi = s.minfo[mbr];
      if ((uint32_t)mi_up(mi) == up) continue;
      s.first_suspect[mbr] = NONE32;
      if (!up) {
        // the process is gone: its piggyback queue with it (member map and deadlines stay: swimsim.h)
        s.crash_tick[mbr] = t;
        // (its inbox with it: a message from outside the simulation may have been delivered before the tick started)
        if (is_local(s, mbr)) { set_mi(s, mbr, mi & ~(MI_UP | MI_PB)); s.pk[mbr - s.lo].x = 0ull; s.inbox_cnt[mbr - s.lo] = 0u; }
        else set_mi(s, mbr, mi & ~MI_UP);
        continue;
      }
      came_up = true;
      if (!is_local(s, mbr)) { set_mi(s, mbr, mi | MI_UP); continue; }   // its owner does the rest
      // (re)join: new incarnation, announce Alive: the queue holds exactly that rumour
      const uint32_t ml = mbr - s.lo;
      const uint2 hot = s.hot[ml];
      uint32_t ni = hot.x + 1;
      if (ni > INC_MAX) { atomicOr(&s.g[G_ERR], (uint32_t)ERRF_INC); ni = INC_MAX; }
      evd += h4(TAG_INC, ((uint64_t)t << 32) | mbr, ni, 0);
      const uint32_t sl = get_slot(s, mbr);
      if (s.G) s.slot_last[sl] = t;
      mi = s.minfo[mbr];
      const uint32_t cur = mi_buf(mi);
      const uint32_t akey = (ni << 2) | ST_ALIVE;
      // a FRESH id, not find_rid's cached one: the mask written below says "the newest id", and the rumour may have been
      // stated before -- a message from outside the simulation can name an incarnation the member has not reached yet
      // (swimsim_inject_rumor; found by the soak: the announcement travelled under an id thousands of ids old, its mask
      // bit read as another rumour).  Two ids for one rumour are harmless (an id is only a filter key).
      const uint32_t arid = new_rid(s) & RID_MASK;
      s.rum[arid] = make_uint2(sl, akey);
      uint64_t* line = s.pb + ((size_t)cur * s.N + ml) * PB_SLOTS;
      line[0] = ((uint64_t)pe_hi(akey, s.L) << 32) | pe_lo(sl, arid);
      for (int q = 1; q < PB_SLOTS; ++q) line[q] = 0ull;
      s.pk[ml] = make_ulonglong2(rid_bit(arid), 0ull);       // its id is the newest: maskable; known-ring empty
      set_mi(s, mbr, (mi & ~(MI_PBN | MI_OOW)) | (1u << MI_PBN_SHIFT) | MI_UP);
      s.inmask[ml] = 0;
      s.hot[ml] = make_uint2(ni, hot.y | 1u);                // merge_kernel fires the deadlines it slept through
      if (s.event_mask & (1u << 4)) {
        uint32_t pos = atomicAdd(&s.g[G_EVCUR], 1u);
        if (pos < s.event_cap) s.events[pos] = make_uint4(t, mbr, mbr, (akey << 8) | 4u);
        else dropped++;
      }
    }
    if (came_up && s.join_pull) joined[atomicAdd(&s.g[G_NJOINED], 1u)] = faults[k0].member;   // whatever it did afterwards
    if (evd) atomicAdd(&evd_sh, evd);
    if (dropped) atomicAdd(&dropped_sh, dropped);
  }
  __syncthreads();
  if (!(part & 2u)) {                               // the rest follows the pulls (part 2)
    if (threadIdx.x == 0) {
      if (evd_sh) s.blk[(size_t)s.nblocks * C_COUNT + C_EVDIGEST] += evd_sh;
      if (dropped_sh) s.blk[(size_t)s.nblocks * C_COUNT + C_EVENTS_DROPPED] += dropped_sh;
    }
    return;
  }
  if (s.join_pull || (s.pull_T && s.n_shards > 1)) {
    // Join-time state pull (`joinHosts`, src/Types.hs:47; include/swimsim.h), behind the barrier: every row this
    // tick's joins opened is complete, and nothing else in this kernel writes view cells.  A member that came up
    // during the tick merges its host's member map: (i) hosts on other shards sent theirs as records {joiner,
    // subject, entry} (rows are opened for subjects this shard has not heard of); (ii) hosts on this shard are read.
    for (uint32_t p = 0; p < s.n_shards; ++p) {
      const uint32_t n = s.n_shards > 1 && p != s.shard ? min(jv.direct ? *jv.jn[p] : jc.v[p], s.j_cap) : 0u;
      const uint4* jlist = jv.direct ? jv.jl[p] : s.j_recv + (size_t)p * s.j_cap;     // (swimsim_cluster_step: the peer's records where they lie)
      for (uint32_t k = threadIdx.x; k < n; k += blockDim.x) {
        const uint4 r = jlist[k];
        if (r.x & PUSH_REC) {
          // the push half of a push-pull whose puller lives on another shard: my member (the host) merges the entry
          const uint32_t host = r.x & ~PUSH_REC;
          if (!is_local(s, host) || r.y >= s.NT) continue;
          unsigned long long evd = 0; unsigned pushed = 0, suspects = 0;
          push_entry(s, t, host, get_slot(s, r.y), r.y, r.z, &evd, &pushed, &suspects);
          if (suspects) atomicOr(&s.hot[host - s.lo].y, 1u);   // a host that took a Suspect over: merge_kernel rebuilds its deadline cells
          if (evd) atomicAdd(&evd_sh, evd);
          if (pushed) atomicAdd(&changes_sh, pushed);
          continue;
        }
        if (!is_local(s, r.x) || r.y >= s.NT) continue;
        unsigned long long evd = 0; unsigned pulled = 0;
        pull_entry(s, t, r.x, get_slot(s, r.y), r.y, r.z, &evd, &pulled);
        // (a periodic puller that took a Suspect over: merge_kernel rebuilds its deadline cells, as join_pull_kernel arranges for
        // the pulls from local hosts; a joiner carries the flag already)
        if (pulled && (r.z & 3u) == ST_SUSPECT) atomicOr(&s.hot[r.x - s.lo].y, 1u);
        if (evd) atomicAdd(&evd_sh, evd);
        if (pulled) atomicAdd(&changes_sh, pulled);
      }
    }
    __syncthreads();
    const uint32_t nj = (part & 8u) ? 0u                     // join_pull_kernel has done (ii): a block per joiner
                                    : atomicOr(&s.g[G_NJOINED], 0u);   // counted with atomics by other waves of this block: read where they landed
    for (uint32_t k = threadIdx.x; k < nj; k += blockDim.x) {
      const uint32_t mbr = joined[k];
      if (!is_local(s, mbr)) continue;
      const uint32_t host = pull_host(s, t, tk, mbr, faults, nfaults, P_JOIN);
      if (host == NONE32 || !is_local(s, host)) continue;
      unsigned long long evd = 0; unsigned pulled = 0;
      const uint32_t nrows = min(s.g[G_NSLOTS], s.R_phys), hl = host - s.lo;
      for (uint32_t r = 0; r < nrows; ++r) {
        if (!s.slot_used[r]) continue;
        const uint32_t subject = s.subject_of[r];
        if (subject == mbr) continue;
        const uint32_t vh = v_key(s, vidx(s, hl, r));
        const uint32_t kh = subject == host ? ((s.hot[hl].x << 2) | ST_ALIVE) : vh;   // an untouched cell is the base: no news
        if (kh) pull_entry(s, t, mbr, r, subject, kh, &evd, &pulled);
      }
      if (evd) atomicAdd(&evd_sh, evd);
      if (pulled) atomicAdd(&changes_sh, pulled);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (evd_sh) s.blk[(size_t)s.nblocks * C_COUNT + C_EVDIGEST] += evd_sh;
    if (dropped_sh) s.blk[(size_t)s.nblocks * C_COUNT + C_EVENTS_DROPPED] += dropped_sh;
    if (changes_sh) s.blk[(size_t)s.nblocks * C_COUNT + C_CHANGES] += changes_sh;
    s.g[G_PREV] = s.g[G_HEAD];
    s.g[G_HEAD] = s.g[G_NRUM];
    s.g[G_ANYREC] = (part & 4u) ? t + 1u : 0u;   // = t + 1, set by whoever writes an explicit record (part bit 2: inject_kernel already has)
    // a line is rewritten every tick and replaces ids outside [H - KW_BITS, H + RID_NEAR) by "no id"; an id born at
    // distance r < RID_NEAR above the head sits at r - D one tick later (D = ids of the tick) and would wrap
    // back INTO that zone for D > 2^RID_BITS - RID_NEAR - KW_BITS.  After such a tick (48 896 new rumours at once with
    // 16-bit ids: heavy message loss) every line of this tick is written without ids: masks and the known-ring
    // are out of the game anyway (explicit records), exactness does not depend on them
    // (strict reference rules: always -- no filter may drop a delivery, see probe_kernel)
    s.g[G_RIDS_OFF] = (s.strict || s.g[G_HEAD] - s.g[G_PREV] > RID_MASK + 1u - RID_NEAR - KW_BITS) ? 1u : 0u;
  }
  for (uint32_t k = threadIdx.x; k <= TODO_REGIONS; k += blockDim.x) s.todo_n[k * 16u] = 0;   // the todo buffer's regions + the spill area
  if (t)                                                            // the deadline chains tick t-1 consumed
    for (uint32_t k = threadIdx.x; k < s.tovf_nsub; k += blockDim.x)
      s.tovf_n[((((t - 1u) % s.S) * 2u + ((((t - 1u) / s.S) & 1u) ^ 1u)) * s.tovf_nsub + k) * 16u] = 0;
  __syncthreads();
  if (threadIdx.x < KN_BITS) {
    // this tick's ring: what each mask / known-ring position stands for, with its row's base and subject, so
    // that merge_kernel resolves a delivered bit from LDS instead of three dependent gathers
    const uint2 r = s.rum[rid_at(threadIdx.x, s.g[G_HEAD]) & RID_MASK];
    const uint32_t row = r.x < s.R_phys ? r.x : 0u;           // a position no id has owned yet: never looked at
    s.ring[threadIdx.x] = make_uint4(r.x, r.y, s.slot_base[row], s.subject_of[row]);
  }
  if (s.G) {
    // rows whose subject nobody has changed its mind about for G ticks (counting this one, checked again by
    // settle_finish): this tick's merge reduces their entries
    const uint32_t nrows = min(s.g[G_NSLOTS], s.R_phys);
    for (uint32_t r = threadIdx.x; r < nrows; r += blockDim.x) {
      if (!s.slot_used[r]) continue;
      const uint32_t last = s.slot_last[r];
      if (last != NONE32 && t - last < s.G) continue;
      const uint32_t k = atomicAdd(&nset, 1u);
      s.settle_slots[k] = r; s.settle_key[k] = 0;
    }
    __syncthreads();
    if (threadIdx.x == 0) { s.g[G_SETTLE_N] = nset; s.g[G_SETTLE_TICK] = t; s.g[G_SETTLE_PENDING] = 1; }
  }
  if (s.n_shards > 1) {
    const uint32_t H = s.g[G_NRUM];
    if (threadIdx.x < 3u * MAX_SHARDS) s.send_cnt[threadIdx.x] = 0;
    if (threadIdx.x == 0) { s.g[G_FLDYN] = 0; s.g[G_XLINES] = 0; }
    // this tick's dictionary for the peers: ring position -> {subject, key} of the id that owns it (one thread per position:
    // a single thread walking the 64 positions made three dependent loads each -- ~100 us of every sharded tick)
    if (threadIdx.x < DICT_ENTRIES) {
      const uint32_t p = threadIdx.x;
      const uint32_t rid = rid_at(p, H);
      uint2 e = make_uint2(NONE32, 0u);
      if (rid < H) {                                   // else: no such id yet
        const uint2 r = s.rum[rid & RID_MASK];
        // with settling a row changes hands: an id handed out before the row went to its present subject names a
        // rumour about the previous one (no queue holds it any more: G >= S + L + 2) -- not in the dictionary, or
        // the peers would open rows for subjects nobody talks about
        const bool live = !s.G || (r.x < s.R_phys && s.slot_used[r.x] && (int32_t)(rid - s.slot_born[r.x]) >= 0);
        if (live) e = make_uint2(s.subject_of[r.x], r.y);
      }
      reinterpret_cast<uint2*>(s.r_send)[p] = e;     // one segment, the same for every peer
    }
  }
}

// The tick's state pulls from hosts on this shard, one BLOCK per puller, between the two parts of begin_kernel (which then
// skips its own loop over the joiners: part bit 3).  Work items [0, nj_bound): the members that came up (`joined`, as many
// as G_NJOINED says); then, with periodic pulls on, the members i = t mod T (mod T).  A puller walks every row of its
// host: one thread per joiner inside begin_kernel's single block made a tick with joins cost 2 dependent gathers x rows,
// serially (5 ms for 25 joiners over 8 000 rows, profiles/r03zz_churn.txt); here the rows go over the block's threads,
// four in flight per thread.  Hosts are members without a change this tick that do not pull in it: the blocks touch
// disjoint members.
__global__ __launch_bounds__(BLOCK) void join_pull_kernel(DevState s, uint32_t t, uint32_t tk, const FaultRec* faults, uint32_t nfaults,
                                                          const uint32_t* joined, uint32_t nj_bound) {
  __shared__ unsigned long long evd_sh;
  __shared__ unsigned pulled_sh, suspects_sh;
  __shared__ uint32_t host_sh;
  const uint32_t nj = s.join_pull ? min(s.g[G_NJOINED], nj_bound) : 0u;
  const uint32_t T = s.pull_T, first = T ? (t % T + T - s.lo % T) % T : 0u;   // my first periodic puller (local index; a shard starts at lo)
  const uint32_t npp = (T && first < s.N) ? (s.N - first + T - 1u) / T : 0u;
  const uint32_t nrows = min(s.g[G_NSLOTS], s.R_phys);
  constexpr int U = 4;
  for (uint32_t k = blockIdx.x; k < nj_bound + npp; k += gridDim.x) {
    const bool joiner = k < nj_bound;
    if (joiner && k >= nj) continue;                           // (block-uniform)
    const uint32_t mbr = joiner ? joined[k] : s.lo + first + (k - nj_bound) * T;
    if (threadIdx.x == 0) {
      evd_sh = 0; pulled_sh = 0; suspects_sh = 0;
      uint32_t h = NONE32;
      if (joiner) { if (is_local(s, mbr)) h = pull_host(s, t, tk, mbr, faults, nfaults, P_JOIN); }
      else if (mi_up(s.minfo[mbr]) && !changes_this_tick(faults, nfaults, mbr)) h = pull_host(s, t, tk, mbr, faults, nfaults, P_PULL);
      host_sh = h;
    }
    __syncthreads();
    const uint32_t host = host_sh;
    if (host != NONE32 && is_local(s, host)) {
      unsigned long long evd = 0; unsigned pulled = 0, suspects = 0;
      const uint32_t hl = host - s.lo;
      const uint32_t hkey = (s.hot[hl].x << 2) | ST_ALIVE;
      for (uint32_t r0 = threadIdx.x; r0 < nrows; r0 += BLOCK * U) {
        uint32_t used[U], subj[U], vh[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const uint32_t r = r0 + u * BLOCK;
          used[u] = r < nrows ? s.slot_used[r] : 0u;
          subj[u] = r < nrows ? s.subject_of[r] : 0u;
          vh[u] = r < nrows ? v_key(s, vidx(s, hl, r)) : 0u;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (!used[u] || subj[u] == mbr) continue;
          const uint32_t kh = subj[u] == host ? hkey : vh[u];   // an untouched cell is the base: no news
          if (!kh) continue;
          const unsigned before = pulled;
          pull_entry(s, t, mbr, r0 + u * BLOCK, subj[u], kh, &evd, &pulled);
          suspects += (pulled != before && (kh & 3u) == ST_SUSPECT) ? 1u : 0u;
        }
      }
      if (evd) atomicAdd(&evd_sh, evd);
      if (pulled) atomicAdd(&pulled_sh, pulled);
      if (suspects) atomicAdd(&suspects_sh, suspects);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      if (evd_sh) atomicAdd(reinterpret_cast<unsigned long long*>(&s.blk[(size_t)s.nblocks * C_COUNT + C_EVDIGEST]), evd_sh);
      if (pulled_sh) atomicAdd(reinterpret_cast<unsigned long long*>(&s.blk[(size_t)s.nblocks * C_COUNT + C_CHANGES]), (unsigned long long)pulled_sh);
      // a periodic puller that took a Suspect over: merge_kernel rebuilds its deadline cells from its view cells, as for a
      // member that came back up (whose flag begin_kernel has set)
      if (!joiner && suspects_sh) s.hot[mbr - s.lo].y |= 1u;
    }
    __syncthreads();
  }
}

// The push half of a push-pull (push_pull = 1; include/swimsim.h "Periodic state pull"): after join_pull_kernel -- every pull has read
// its host -- the host of every periodic puller merges the puller's map: one block per puller, the rows over its threads.  A host may
// have several pullers (several blocks write its cells): the key is raised with atomicMax, whoever raises it accounts for the step it
// made (the event digest is linear in the key: the steps telescope to final - initial whatever their order), the first to stamp
// lastChange counts the change.  Hosts are never pullers: nobody reads a map that is written here.
__global__ __launch_bounds__(BLOCK) void push_kernel(DevState s, uint32_t t, uint32_t tk, const FaultRec* faults, uint32_t nfaults) {
  __shared__ unsigned long long evd_sh;
  __shared__ unsigned pushed_sh, suspects_sh;
  __shared__ uint32_t host_sh;
  const uint32_t T = s.pull_T, first = (t % T + T - s.lo % T) % T;   // my first periodic puller (local index; a shard starts at lo)
  const uint32_t npp = first < s.N ? (s.N - first + T - 1u) / T : 0u;
  const uint32_t nrows = min(s.g[G_NSLOTS], s.R_phys);
  constexpr int U = 4;
  for (uint32_t k = blockIdx.x; k < npp; k += gridDim.x) {
    const uint32_t mbr = s.lo + first + k * T;
    if (threadIdx.x == 0) {
      evd_sh = 0; pushed_sh = 0; suspects_sh = 0;
      uint32_t hst = (mi_up(s.minfo[mbr]) && !changes_this_tick(faults, nfaults, mbr)) ? pull_host(s, t, tk, mbr, faults, nfaults, P_PULL) : NONE32;
      // (a host on another shard takes my map as records: pull_send_kernel wrote them, its begin_kernel merges them)
      if (hst != NONE32 && !is_local(s, hst)) hst = NONE32;
      host_sh = hst;
    }
    __syncthreads();
    const uint32_t host = host_sh;
    if (host != NONE32) {
      unsigned long long evd = 0; unsigned pushed = 0, suspects = 0;
      const uint32_t ml = mbr - s.lo, hl = host - s.lo;
      const uint32_t mkey = (s.hot[ml].x << 2) | ST_ALIVE;
      const unsigned long long hw = mix64(mix64((uint64_t)TAG_EV) + (((uint64_t)t << 32) | host));
      for (uint32_t r0 = threadIdx.x; r0 < nrows; r0 += BLOCK * U) {
        uint32_t used[U], subj[U], vm[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const uint32_t r = r0 + u * BLOCK;
          used[u] = r < nrows ? s.slot_used[r] : 0u;
          subj[u] = r < nrows ? s.subject_of[r] : 0u;
          vm[u] = r < nrows ? v_key(s, vidx(s, ml, r)) : 0u;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (!used[u] || subj[u] == host) continue;
          const uint32_t r = r0 + u * BLOCK;
          const uint32_t km = subj[u] == mbr ? mkey : vm[u];     // an untouched cell is the base: no news
          if (!km || km <= s.slot_base[r]) continue;              // (an untouched cell of the host reads 0: never raise it to below the base)
          const size_t ix = vidx(s, hl, r);
          const uint32_t old = v_raise_key(s, ix, km, t);
          const uint32_t curk = old ? old : s.slot_base[r];
          if (km <= curk) continue;
          evd += (mix64(hw + subj[u]) | 1ull) * (unsigned long long)(km - curk);
          if (v_stamp(s, ix, t) != t + 1u) pushed++;
          if (s.G) s.slot_last[r] = t;
          suspects += (km & 3u) == ST_SUSPECT ? 1u : 0u;
        }
      }
      if (evd) atomicAdd(&evd_sh, evd);
      if (pushed) atomicAdd(&pushed_sh, pushed);
      if (suspects) atomicAdd(&suspects_sh, suspects);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      if (evd_sh) atomicAdd(reinterpret_cast<unsigned long long*>(&s.blk[(size_t)s.nblocks * C_COUNT + C_EVDIGEST]), evd_sh);
      if (pushed_sh) atomicAdd(reinterpret_cast<unsigned long long*>(&s.blk[(size_t)s.nblocks * C_COUNT + C_CHANGES]), (unsigned long long)pushed_sh);
      // a host that took a Suspect over: merge_kernel rebuilds its deadline cells from its view cells (as for a puller)
      if (suspects_sh) atomicOr(&s.hot[host - s.lo].y, 1u);
    }
    __syncthreads();
  }
}

// Sharded clusters with join_pull: between the two parts of begin_kernel the owner of a join host sends what the host
// knows to the joiner's owner -- one record {joiner, subject, entry} per entry that differs from the base (the host
// itself as Alive at its own incarnation) -- exchange round 0.  One thread per member that came up this tick.
__global__ __launch_bounds__(BLOCK) void pull_send_kernel(DevState s, uint32_t t, uint32_t tk, const FaultRec* faults, uint32_t nfaults,
                                                           const uint32_t* joined) {
  // work items: the members that came up in this tick (join_pull), then -- pull_ticks = T -- the periodic pullers of the WHOLE population,
  // t mod T, t mod T + T, ...: every shard looks at all of them and serves those whose host it owns and whose puller it does not
  const uint32_t nj = s.join_pull ? s.g[G_NJOINED] : 0u;
  const uint32_t T = s.pull_T, first = T ? t % T : 0u;
  const uint32_t npp = (T && first < s.NT) ? (s.NT - first + T - 1u) / T : 0u;
  for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < nj + npp; k += gridDim.x * blockDim.x) {
    const bool joiner = k < nj;
    const uint32_t mbr = joiner ? joined[k] : first + (k - nj) * T;
    if (is_local(s, mbr)) {
      // push_pull (round 6): a periodic puller of MINE whose host lives elsewhere hands the host's owner its map -- one record
      // {host | PUSH_REC, subject, my entry} per entry that differs from the base, itself as Alive at its own incarnation.  The map
      // BEFORE the pull: the host ends at max(its map, mine) either way (the pull adds nothing to my map the host does not hold)
      if (joiner || !s.push_pull || !mi_up(s.minfo[mbr]) || changes_this_tick(faults, nfaults, mbr)) continue;
      const uint32_t host = pull_host(s, t, tk, mbr, faults, nfaults, P_PULL);
      if (host == NONE32 || is_local(s, host)) continue;
      const uint32_t peer = owner_of(s, host), nrows = min(s.g[G_NSLOTS], s.R_phys), ml = mbr - s.lo;
      for (uint32_t r = 0; r < nrows; ++r) {
        if (!s.slot_used[r]) continue;
        const uint32_t subject = s.subject_of[r];
        if (subject == host) continue;
        const uint32_t km = subject == mbr ? ((s.hot[ml].x << 2) | ST_ALIVE) : v_key(s, vidx(s, ml, r));
        if (!km) continue;
        const uint32_t pos = atomicAdd(&s.g[G_JSEND + peer], 1u);
        if (pos < s.j_cap) s.j_send[(size_t)peer * s.j_cap + pos] = make_uint4(host | PUSH_REC, subject, km, 0u);
        else atomicOr(&s.g[G_ERR], (uint32_t)ERRF_XCHG);
      }
      continue;
    }
    if (!joiner && (!mi_up(s.minfo[mbr]) || changes_this_tick(faults, nfaults, mbr))) continue;   // (a puller is up and has no change in this tick)
    const uint32_t host = joiner ? join_host(s, t, tk, mbr, faults, nfaults) : pull_host(s, t, tk, mbr, faults, nfaults, P_PULL);
    if (host == NONE32 || !is_local(s, host)) continue;
    const uint32_t peer = owner_of(s, mbr), nrows = min(s.g[G_NSLOTS], s.R_phys), hl = host - s.lo;
    for (uint32_t r = 0; r < nrows; ++r) {
      if (!s.slot_used[r]) continue;
      const uint32_t subject = s.subject_of[r];
      if (subject == mbr) continue;
      const uint32_t kh = subject == host ? ((s.hot[hl].x << 2) | ST_ALIVE) : v_key(s, vidx(s, hl, r));
      if (!kh) continue;
      const uint32_t pos = atomicAdd(&s.g[G_JSEND + peer], 1u);
      if (pos < s.j_cap) s.j_send[(size_t)peer * s.j_cap + pos] = make_uint4(mbr, subject, kh, 0u);
      else atomicOr(&s.g[G_ERR], (uint32_t)ERRF_XCHG);
    }
  }
}

// Rumours from outside the simulation (swimsim_inject_rumor; src/Core.hs:110-117 for a message off the socket): each
// becomes a one-entry "foreign line" and an explicit record in its observer's inbox -- the path payloads from other
// shards take -- so that this tick's merge rules on it next to everything else the member received.  After
// begin_kernel (the tick's window head is fixed: an id handed out here is younger than it and travels unfiltered).
struct InjectRec { uint32_t observer, subject, key, pad; };
__global__ void inject_kernel(DevState s, uint32_t t, const InjectRec* recs, uint32_t n) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const InjectRec r = recs[k];
  if (!mi_up(s.minfo[r.observer])) return;          // nobody listening
  const uint32_t slot = get_slot(s, r.subject);
  // (a shard that does not own the observer: the subject's view row only -- a row is a property of the whole cluster, DESIGN.md 2.4 /
  // 2.6: a state pull on THIS shard walks it in this very tick)
  if (r.pad) return;
  uint32_t num = 0;
  (void)num;
  const uint32_t rid = find_rid(s, slot, r.key, &num);      // handed out before the tick's window head is taken: an id like any of the tick before
  uint4* line = s.fl + ((size_t)s.fl_inj_base + k) * 4;
  line[0] = make_uint4(pe_lo(slot, rid), pe_hi(r.key, 1u), 0u, 0u);
  line[1] = make_uint4(0u, 0u, 0u, 0u); line[2] = make_uint4(0u, 0u, 0u, 0u); line[3] = make_uint4(0u, 0u, 0u, 0u);
  __threadfence();
  push(s, t, r.observer - s.lo, SRC_FOREIGN | (s.fl_inj_base + k));
}

// full-state digest: Sum_i mix64(member_hash(i) + mix64(TAG_MEMBER + i)) + first-detection terms
// swimsim_coverage: how many of my up members (the subject apart) hold an entry about `subject` that is at least `key`
// in merge order -- out[0]; how many such members there are -- out[1].  One coalesced walk along the subject's row.
__global__ __launch_bounds__(BLOCK) void coverage_kernel(DevState s, uint32_t subject, uint32_t key, unsigned long long* out) {
  const uint32_t li = blockIdx.x * BLOCK + threadIdx.x;
  const uint32_t sl1 = s.minfo[subject] & MI_SLOT;           // 0: nobody here has heard of it (no row)
  const uint32_t base = sl1 ? s.slot_base[sl1 - 1u] : s.base_key[subject];
  uint32_t up = 0, hold = 0;
  if (li < s.N && s.lo + li != subject && mi_up(s.minfo[s.lo + li])) {
    up = 1;
    const uint32_t k = sl1 ? v_key(s, vidx(s, li, sl1 - 1u)) : 0u;
    hold = (k ? k : base) >= key ? 1u : 0u;                   // an untouched cell is the settled base
  }
  const unsigned long long bh = __ballot(hold != 0u), bu = __ballot(up != 0u);
  if ((threadIdx.x & 63u) == 0u) {
    if (bh) atomicAdd(&out[0], (unsigned long long)__popcll(bh));
    if (bu) atomicAdd(&out[1], (unsigned long long)__popcll(bu));
  }
}

__global__ __launch_bounds__(BLOCK) void digest_kernel(DevState s, unsigned long long* out) {
  __shared__ unsigned long long acc;
  if (threadIdx.x == 0) acc = 0;
  __syncthreads();
  const uint32_t li = blockIdx.x * BLOCK + threadIdx.x;
  const uint32_t i = s.lo + li;
  if (li < s.N) {
    const uint2 hot = s.hot[li];
    const uint32_t mi = s.minfo[i];
    unsigned long long mh = h4(TAG_SELF, i, hot.x, mi_up(mi) ? 1u : 0u);
    const uint32_t ns = min(s.g[G_NSLOTS], s.R_phys);
    for (uint32_t r = 0; r < ns; ++r) {
      if (!s.slot_used[r]) continue;                 // reclaimed (its cells are cleared by the next merge)
      const uint2 e = v_full(s, vidx(s, li, r));
      if (e.x == 0) continue;
      const uint32_t subject = s.subject_of[r];
      if (subject == i) continue;
      mh += h4(TAG_VIEW, subject, e.x, e.y);
      if ((e.x & 3u) == ST_SUSPECT) mh += h4(TAG_TIMER, subject, (uint64_t)e.y - 1 + s.S, 0);
    }
    if (mi_pbn(mi)) {
      const uint64_t* line = s.pb + ((size_t)mi_buf(mi) * s.N + li) * PB_SLOTS;
      for (int q = 0; q < PB_SLOTS; ++q) {
        const uint32_t lo = (uint32_t)line[q], hi = (uint32_t)(line[q] >> 32);
        if (pe_tx(hi)) mh += h4(TAG_PB, s.subject_of[pe_slot(lo)], pe_key(hi), pe_tx(hi));
      }
    }
    unsigned long long d = mix64(mh + mix64((uint64_t)TAG_MEMBER + i));
    const uint32_t fs = s.first_suspect[i];
    if (fs != NONE32) d += h4(TAG_FD, i, fs, 0);
    const uint32_t bk = s.base_key[i];
    if (bk) d += h4(TAG_BASE, i, bk, s.base_since[i]);
    atomicAdd(&acc, d);
  }
  __syncthreads();
  if (threadIdx.x == 0 && acc) atomicAdd(out, acc);
}

// kRandomMembers for one observer (unit-level hook; test/Spec.hs:108-139)
__global__ void select_debug_kernel(DevState s, uint32_t tk, uint32_t observer, uint32_t n,
                                    const uint32_t* excl, uint32_t nexcl, uint32_t* out, uint32_t* n_out) {
  if (blockIdx.x || threadIdx.x) return;
  uint32_t picks[256], info[256];
  const uint32_t np = select_members<256>(s, mix32(tk ^ observer), observer, n, P_SELECT, 0, excl, nexcl, picks, info);
  for (uint32_t k = 0; k < np; ++k) out[k] = picks[k];
  *n_out = np;
}

// overwrite one view entry (test fixture hook)
__global__ void set_view_kernel(DevState s, uint32_t t, uint32_t observer, uint32_t subject, uint32_t key) {
  if (blockIdx.x || threadIdx.x) return;
  ensure_slot(s, subject);
  const uint32_t sl = (s.minfo[subject] & MI_SLOT) - 1;
  const uint32_t ol = observer - s.lo;
  v_put(s, vidx(s, ol, sl), key, t + 1);
  if ((key & 3u) == ST_SUSPECT) {                   // deadline t + S: row t mod S (merge_kernel carries it over)
    const size_t ix = (size_t)(t % s.S) * s.N + ol;
    const uint4 cell = s.trow[ix];
    TimerCell c; c.lo = cell.x | ((unsigned long long)cell.y << 32); c.hi = cell.z | ((unsigned long long)cell.w << 32); c.n = 0;
    while (c.n < TR_PAY && tc_get(cell, c.n)) c.n++;
    tc_put_simple(c, sl + 1);
    s.trow[ix] = tc_pack(c);
  }
}

__global__ void init_members_kernel(uint32_t* minfo, uint8_t* mb, uint32_t n_total) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_total) { minfo[i] = MI_UP; mb[i] = (uint8_t)mb_of(MI_UP); }
}

}  // namespace swim
