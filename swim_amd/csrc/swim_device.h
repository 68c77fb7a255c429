// swim_device.h -- device-side data layout and spec arithmetic for the SWIM tick (gfx950).
//
// Spec: DESIGN.md section 2.  Reference rules cited per function (paths relative to the
// jpfuentes2/swim checkout).  This header is the PRODUCT's own statement of the spec hashes
// and packings; the CPU oracle under oracle/ restates them independently.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace swim {

// ---- spec constants -------------------------------------------------------------------
constexpr int PB_SLOTS = 8;           // piggyback buffer slots per member (D5): one 64-B line
constexpr int SEL_ATTEMPTS = 8;       // rejection-sampling attempts per pick (H4)
constexpr uint32_t INC_MAX = 0x3FFFFFu;
constexpr uint32_t NONE32 = 0xFFFFFFFFu;
constexpr int BLOCK = 256;

enum { ST_ALIVE = 0, ST_SUSPECT = 1, ST_DEAD = 2 };
enum { P_SELECT = 1, P_PROXY = 2, P_L_PING = 3, P_L_ACK = 4, P_L_REQ = 5, P_L_FWD = 6, P_L_BACK = 7,
       P_L_RELAY = 8, P_JOIN = 9, P_PULL = 10 };
enum : uint64_t { TAG_SELF = 0x53454c46u, TAG_VIEW = 0x56494557u, TAG_PB = 0x50425546u,
                  TAG_TIMER = 0x54494d52u, TAG_FD = 0x46444554u, TAG_EV = 0x45564e54u,
                  TAG_INC = 0x494e4352u, TAG_TICK = 0x5449434bu, TAG_MEMBER = 0x4d454d42u,
                  TAG_BASE = 0x42415345u };

// globals word indices (DevState::g)
enum { G_NSLOTS = 0 /* view rows ever handed out (high-water mark) */, G_ERR = 1, G_EVCUR = 2, G_OVF0 = 3, G_OVF1 = 4,
       G_NRUM = 5, G_HEAD = 6, G_PREV = 7,
       G_NFREE = 8 /* reclaimed rows on the free stack */, G_NLIVE = 9 /* subjects with a row (max_subjects bounds it) */,
       G_SETTLE_N = 10 /* rows whose entries this tick's merge reduces */, G_ZERO_N = 11 /* rows it clears */,
       G_SETTLE_PENDING = 12 /* the lists above still await settle_finish */, G_SETTLE_TICK = 13,
       G_RIDS_OFF = 14 /* so many new rumours last tick that this tick's lines carry no ids at all */,
       G_SETTLE_SEND = 15 /* sharded settling: records this shard publishes at the end of the tick (same list to every peer) */,
       G_SEND = 16 /* [3][16] exchange records appended per peer (send_cnt) */,
       G_NJOINED = 64 /* members that came up in this tick (begin_kernel part A) */,
       G_JSEND = 65 /* [16] join-pull records appended per peer (exchange round 0) */,
       G_FLDYN = 81 /* foreign lines handed out by this tick's deliveries from other shards */,
       G_XLINES = 82 /* queues published as lists this tick (publish_kernel) */,
       G_ANYREC = 83 /* = t + 1: somebody wrote an explicit record in tick t */,
       G_HEAD_NEW = 84, G_PREV_NEW = 85 /* a tick WITHOUT begin_kernel (swimsim_step's plain ticks): the tick's window head / the one before,
                                           left by probe_kernel's workgroup 0 for merge_kernel, whose workgroup 0 commits them to G_HEAD / G_PREV */,
       G_WORDS = 96 };
// the todo buffer (explicit records' survivors) is cut into TODO_REGIONS regions with a counter each, on 64-byte lines of
// their own (todo_n[region * 16]): workgroup b reserves from region b mod TODO_REGIONS -- every wave of the grid bumping
// ONE word was 12 % of merge_kernel's wave time at 1 % loss (same-address atomics serialise, profiles/r03t_*)
constexpr uint32_t TODO_REGIONS = 64;
enum { ERRF_SUBJECTS = 1, ERRF_ROWS = 2, ERRF_OVF = 4, ERRF_INC = 8, ERRF_XCHG = 16, ERRF_TODO = 32 };

// counter slots (same order as SWIMSIM_CTR_* in include/swimsim.h)
enum { C_PINGS = 0, C_DIRECT_FAILED, C_PING_REQS, C_SUSPECTS, C_FALSE_SUSPECTS, C_PAYLOADS,
       C_RUMORS_SEEN, C_CHANGES, C_PB_WRITES, C_TIMERS_FIRED, C_REFUTES, C_EVENTS_DROPPED,
       C_ACTIVE, C_EVDIGEST, C_FALSE_DEADS /* timers fired about a member that is up */, C_SETTLED,
       C_EVICTED /* bounded member maps (view_cap): entries returned to the default */, C_COUNT = 17 };

// ---- hashes (replace the global StdGen of src/Util.hs:40, F7) -----------
__host__ __device__ inline uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__host__ __device__ inline uint32_t tick_key(uint64_t seed, uint32_t t) {
  return mix32((uint32_t)seed + mix32((uint32_t)(seed >> 32) + mix32(t + 0x9E3779B9u)));
}
// H(tk, a, b, c); mk = mix32(tk ^ a) is hoisted per member
__host__ __device__ inline uint32_t hash_mk(uint32_t mk, uint32_t b, uint32_t c) {
  return mix32(mix32(mk + b) ^ c);
}
__host__ __device__ inline uint64_t mix64(uint64_t x) {
  x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull; x ^= x >> 27; x *= 0x94d049bb133111ebull; x ^= x >> 31;
  return x;
}
__host__ __device__ inline uint64_t h4(uint64_t tag, uint64_t a, uint64_t b, uint64_t c) {
  return mix64(mix64(mix64(mix64(tag) + a) + b) + c);
}

// ---- device state (struct of arrays) -----------------------------------------------------
// Per-member arrays are indexed by member id so that a wave of 64 consecutive members
// issues coalesced loads.  The tick is bound by the NUMBER of L2<->fabric requests (measured:
// ~45-55 G requests/s for scattered 64-B accesses, profiles/), so the two big owner-private tables
// are row-major over members: view entries V[slot][member] and suspicion deadlines
// trow[deadline mod S][member].  The members of a wave look at the same few rumour slots (the ones
// in circulation) -- merge_kernel walks them in wave-uniform order -- and every member reads and
// rewrites the SAME deadline row in a tick, so these accesses share sectors instead of costing one
// request per member.
struct DevState {
  // N = members owned by this handle (local index li in [0,N)), NT = whole population, lo = global id of
  // local member 0 (unsharded: lo = 0, NT = N).  minfo / first_suspect / crash_tick are indexed by GLOBAL
  // id (ground truth and the subject -> slot table are replicated on every shard); every other
  // per-member array by LOCAL index.  Hashes, events and digests always use global ids.
  uint32_t N, NT, lo, n_shards, shard;
  uint32_t scheme;         // SWIMSIM_TARGETS_*: how the direct probes of a period pick their targets
  uint32_t join_pull;      // a member that comes up merges a join host's member map (include/swimsim.h)
  uint32_t pull_T;         // periodic state pull: member i pulls in the ticks t = i (mod pull_T); 0 = off (include/swimsim.h)
  uint32_t* sp_bloom;      // bounded member maps: [N][2^(sp_bloom_log2 - 5)] "not Alive in my view" filters (swim_sparse.h)
  uint32_t sp_bloom_log2;  // bits per filter, log2
  uint32_t fl_inj_base;    // first foreign line of the injected rumours (0 on unsharded handles: they are the only foreign lines)
  uint32_t push_pull;      // the periodic pull's host merges the puller's map too (include/swimsim.h "Periodic state pull")
  uint32_t strict;         // strict_reference_rules: the literal suspectOrDeadNode' under the canonical order (include/swimsim.h; D13)
  uint32_t P, K, S, L, loss_thr, R_max /* max_subjects */, R_phys /* view rows allocated */, G /* settling horizon, 0 = off */;
  uint32_t event_cap, event_mask, nblocks;
  uint32_t inbox_cap, ovf_cap;  // per-member delivery slots; exact overflow list capacity
  uint32_t* minfo;         // per member, ONE gather per probe target:
                           //   bits 0-15 rumour slot+1 of this member as a subject (0 none,
                           //   0xFFFF being allocated), 16-19 valid piggyback slots,
                           //   20 which pb buffer is current, 21 up (ground truth),
                           //   22 the queue holds an entry its mask cannot express (MI_OOW),
                           //   23-24 state of the settled base entry about this member (everybody's default)
  uint8_t* mb;             // [NT] the bits of minfo a prober needs about a target, in ONE byte (a 1-MB table at 2^20
                           //   members stays resident in every XCD's L2, the 4-MB minfo table does not -- the
                           //   gathers are the probe kernel's bound): bit 0 up, 1-4 queue length, 5 MI_OOW,
                           //   6 the member has a view row (then, and on explicit-record ticks, the prober
                           //   reads the full word), 7 its settled base is not Alive.  Written with every minfo.
  uint16_t* probe_out;     // nsent | nfail<<5 | n explicit own-ack sources<<10, probe -> merge kernel
  ulonglong2* pk;          // per member {x: the queue as a 64-bit mask over rumour-id positions (rid & 63),
                           //   y: known-ring, bit (rid & 63) set => this member's view already dominates
                           //   rumour rid}: ONE 16-byte gather per probe target serves the Ack's payload
                           //   (x) and the "anything new for you?" test before a push (y)
  unsigned long long* inmask;   // OR of the masks pushed to this member this tick (atomicOr by the pingers)
  unsigned long long* ackmask;  // OR of the masks this member pulled with its Acks (plain store by the prober)
  uint2* rum;              // [1 << RID_BITS] rumour id -> {slot, key}
  ulonglong4* kw;          // [N] wide known-ring (explicit-record path; above), positions beyond KW_BITS unused
  uint32_t* kw_head;       // [N] head the member's kw was written at
  uint4* ring;             // [64] this tick's ring, built by begin_kernel: position -> {slot, key, row base, subject}
  unsigned long long* rtab;// [R_max][RT_WAYS] (slot, key) -> rumour id: {key+1 : 32 | allocation number >> RID_BITS : 15 | ready : 1 | rid : 16}
  // explicit delivery records "dst merges src's 64-B line": the exact fallback for queues with
  // entries outside the mask window, and for ticks that follow a burst of new rumour ids
  uint32_t* ackfrom;       // [N][P] sources whose Ack reached this member with such a payload
  uint32_t* inbox_cnt;     // explicit deliveries to this member this tick
  uint32_t* inbox;         // [N][inbox_cap] source ids (bit31 = source's pb buffer)
  uint2* todo;             // [todo_cap] what the records phase of merge_kernel leaves for the rest of the kernel: the record entries that survived the rings,
                           //   {slot | rid << 16, key}; a member's list is todo_seg[member], its total
                           //   length replaces inbox_cnt[member]; kn_rec[member] = ring positions of the ids they carried
  uint4* todo_seg;         // [N] a member's list in two segments {start 0, length 0, start 1, length 1} (the flattened pass of
                           //   records_kernel; what its serial pass adds: overflow-list sources, members with more sources than a chunk)
  unsigned long long* kn_rec;
  uint32_t* todo_n;        // [(TODO_REGIONS + 1) * 16] entries handed out per region this tick, the last one: of the spill area (begin_kernel zeroes them)
  uint32_t todo_spill_at, todo_spill;   // the spill area all regions share: first entry, entries
  uint32_t todo_cap;       // entries per region
  uint2* hot;              // {storeIncarnation, flags: bit 0 = came back up, deadlines slept through not fired yet}
  uint32_t* subject_of;    // slot -> subject
  uint32_t* fail;          // [N][P] targets whose probe ended without ack
  uint4* trow;             // [S][N] suspicion deadlines: row d mod S = the rows (slot+1, 16 bit) this member must
                           //   look at in tick d (the FIXME at src/Core.hs:141; D4).  Every member reads and
                           //   rewrites row t mod S in tick t: coalesced, no per-member FIFO.
  uint4* tovf;             // [S][2][tovf_cap] overflow cells of the deadline rows (chains, two pools per row)
  uint32_t* tovf_n;        // [S][2][tovf_nsub][16] cells handed out, one counter (on its own 64-B line) per sub-pool:
                           // block b takes cells from sub-pool b mod tovf_nsub -- tens of thousands of members spill per
                           // tick under message loss, and one counter serialises them (same-address atomics)
  uint32_t tovf_cap, tovf_nsub, tovf_sub_cap;   // cells per (row, parity); sub-pools (power of two); cells per sub-pool
  uint2* V;                // [R_phys][N] {key = inc<<2|state, lastChange+1}; key 0 = default = slot_base[slot]
  // SWIM_VSPLIT (round 5): the same cells as two 4-byte planes -- Vk[row][member] = key << 8 | (lastChange + 1) & 0xFF, the
  // only word the hot path LOADS (16 members per 64-byte sector instead of 8: merge_kernel runs at the chip's random-SECTOR
  // rate), Vs[row][member] = lastChange + 1 in full: stored with every change, loaded by the cold paths
  // (digest, views, the walk of a member that woke up) and where the low byte cannot decide (below)
  uint32_t* Vk; uint32_t* Vs;
  // ---- settling (gc_ticks; include/swimsim.h): removeDeadNodes (src/Core.hs:65-67)
  uint32_t* slot_last;     // [R_phys] last tick any entry of the row changed / its subject announced itself
  uint32_t* slot_base;     // [R_phys] base key of the row's subject (what a cell with key 0 means)
  uint8_t* slot_used;      // [R_phys] 1 = the row belongs to subject_of[row]
  uint32_t* base_key;      // [NT] settled entry about a member (0 = Alive@0), base_since its settling tick
  uint32_t* base_since;
  uint32_t* free_rows;     // [R_phys] stack of reclaimed rows (popped by ensure_slot, pushed by settle_finish)
  uint32_t* settle_slots;  // [R_phys] rows eligible this tick; settle_key[k] = max entry among up members
  uint32_t* settle_key;
  uint32_t* settle_part;   // [R_phys][nblocks] merge_kernel's per-block maxima of the eligible rows (no atomics)
  uint32_t* zero_slots;    // [R_phys] rows settled at the end of the last tick: cleared by this tick's merge
  uint32_t* slot_born;     // [R_phys] value of the id counter when the row was handed to its subject: a rumour id older than
                           //   that names a rumour of the row's PREVIOUS subject (sharded settling: the tick's dictionary)
  uint64_t* pb;            // [2][N][PB_SLOTS] {lo: slot | rid<<16, hi: key | tx<<24}, sorted by priority
  uint32_t* first_suspect;
  uint32_t* crash_tick;
  uint32_t* g;             // globals (G_*)
  uint2* ovf;              // [2][ovf_cap] inbox overflow (dst, src)
  uint4* events;           // {tick, observer, subject, key<<8|cause}
  uint64_t* blk;           // [nblocks+1][C_COUNT] per-block counter rows (no atomics)
  // ---- cross-shard exchange (n_shards > 1) ------------------------------
  // Slots (view rows) and rumour ids are per-shard numberings; what crosses shards is named by (subject, key).  Every shard
  // holds a REPLICA of what a delivery "dst merges src's queue" reads about src -- its start-of-tick queue mask (over its
  // OWNER's ring of the tick) and a queue byte -- all-gathered at the start of the tick together with every shard's ring
  // dictionary (position -> {subject, key}) and the queues a mask cannot express as lists of (subject, key).  A delivery to a
  // member of another shard is then ONE 8-byte record {dst, src} to the owner of dst; everything else -- outcomes, the
  // Acks' payloads, every filter -- is a local read.  Nothing is requested, nothing is answered.
  uint2* ord;              // [nblocks][ord_cap] {dst, src} (global ids): deliveries a probe block hands to a shard's ingest
                           //   (the owner of dst -- possibly this shard itself: a remote source whose queue needs more than a
                           //   mask translation); routed into q_send by the block itself at the end of probe_kernel
  uint32_t ord_cap;
  uint4* r_send; uint4* r_recv;         // round 1, 16-byte records: [DICT_RECS + r_cap] / [n_shards][DICT_RECS + r_cap]: this
                                        //   tick's ring dictionary (64 x {subject, key}), then the queues that travel as lists
                                        //   (XLINE_RECS records each: {member, n, tick, -} + 8 x {subject, key}); ONE segment
                                        //   goes to every peer
  uint4* p_send; uint4* p_recv;         // bounded handles (swim_sparse.h): [n_shards][p_cap] {dst, src, -, -}
  uint2* q_send; uint2* q_recv;         // round 2: [n_shards][p_cap] {dst, src}; segment [shard] of q_send stays here
  uint32_t r_cap, p_cap, x_cap;
  uint32_t* send_cnt;      // = g + G_SEND: [3][MAX_SHARDS] records appended per peer ([0][0]: round-1 records, [1][p]: round-2
                           //   records for peer p) -- inside g so that ONE small copy brings flags and counts to the host
  uint2* xl;               // [n_shards][64] a peer's ring dictionary in MY numbering {slot | rid<<16, key | my ring position<<24
                           //   (0xFF: my masks cannot carry it this tick)}; NONE32 = the peer's position is empty
  uint32_t* xidx;          // [NT] where a remote member's list (r_recv) lies, as of the list's own tick stamp
  uint4* fl;               // "foreign lines": received entries my masks cannot carry, read through explicit records
  // sharded settling: what every shard says about its rows at the end of a tick, all-gathered
  uint2* s_send; uint2* s_recv;         // [n_shards][s_cap] {subject | SR_CAND / SR_VETO, largest entry among my up members}
  uint32_t s_cap;
  uint32_t* settle_acc;    // [NT] scratch of settle_commit_kernel (zero between launches)
  // join-time pulls whose host lives on another shard (round 0): {joiner, subject, the host's entry, -}
  uint4* j_send; uint4* j_recv;         // [n_shards][j_cap]
  uint32_t j_cap;
  unsigned long long* mask_all;  // [NT] start-of-tick queue mask over the OWNER's ring of the tick (own slice: publish_kernel)
  uint8_t* q_all;                // [NT] bits 0-3 queue length, 4 = the queue travels as a list this tick (Q_OOW: an entry the mask
                                 //   cannot express, or a tick in which the owner's masks are off)
  uint32_t fl_dyn_base, fl_dyn_cap;   // region of `fl` handed out by the tick's deliveries (G_FLDYN)
  // ---- bounded member maps (view_cap = C > 0; swim_sparse.h): none of the view / mask / deadline tables above
  uint32_t C;              // entries a member's map holds at most; 0 = the unbounded layout above
  uint32_t* sp_tab;        // [N][3][C] the maps: subjects, keys, lastChange + 1 (three coalesced runs per member)
  uint32_t* sp_tab_n;      // [N] entries in use
  uint2* sp_q;             // [2][N][8] queue lines {subject, key | tx << 24}; buffer (t & 1) is read in tick t, the other written
  uint32_t* sp_out;        // [N] probe -> merge: Pings sent | failed probes << 5 | own Ack sources << 10
  uint32_t sp_ack_cap;     // own Ack sources per member: P (1 + K), in ackfrom[N][sp_ack_cap]
  // shards of a bounded cluster (swim_sparse.h): a replica of everybody's start-of-tick queue line (all-gathered with
  // `mb` at the start of the tick), the tick's deliveries to members of other shards
  uint2* sp_qall;          // [NT][8]
  uint2* sp_ord;           // [64][sp_ord_cap] {dst, src} (global ids), unsorted; sp_ord_n[64 * 16] entries per list
  uint32_t* sp_ord_n;
  uint32_t sp_ord_cap;
  uint32_t* sp_pin;        // [MAX_SHARDS] records received from every peer this tick (swimsim_cluster_step: device-side counts)
#ifdef SWIM_ABLATE
  uint32_t dbg;            // measurement build (scripts/ablate.py): memory operations the tick kernels leave out
#endif
};
// -DSWIM_ABLATE (libswimsim_abl.so, measurement only: results are WRONG by construction): a tick kernel skips the class
// of memory operations named by a bit of DevState::dbg, so that its share of the launch time can be read off
#ifdef SWIM_ABLATE
#define ABL(bit) ((s.dbg & (uint32_t)(bit)) != 0u)
#else
#define ABL(bit) false
#endif
enum { ABL_PUSH_ATOMIC = 1, ABL_PK_GATHER = 2, ABL_MB_GATHER = 4, ABL_ACKMASK_STORE = 8, ABL_V_STORE = 16, ABL_V_LOAD = 32,
       ABL_OWN_LINE = 64, ABL_LINE_STORE = 128, ABL_EVD = 256, ABL_FIND_RID = 512, ABL_GROUP = 1024, ABL_STATE_STORES = 2048,
       ABL_INPUTS = 4096, ABL_DEADLINES = 8192, ABL_RUMOURS = 16384 };
constexpr uint32_t Q_PBN = 0xFu, Q_OOW = 1u << 4;
// settle records: a shard lists a row as a CANDIDATE (quiet here for G ticks; y = the largest entry among its members
// that are up) or as a VETO (an entry changed / the subject announced itself within the last G ticks)
constexpr uint32_t SR_CAND = 1u << 30, SR_VETO = 1u << 31, SR_KEY = 0xFFFFFFu;
constexpr uint32_t DICT_ENTRIES = 64;   // one dictionary entry per ring position ...
constexpr uint32_t DICT_RECS = 32;      // ... = 32 sixteen-byte records at the head of every round-1 segment
constexpr uint32_t XLINE_RECS = 5;      // a queue as a list: {member, n, tick, -} + PB_SLOTS x {subject, key} = 80 bytes
constexpr uint32_t ID_BITS = 27, ID_MASK = (1u << ID_BITS) - 1u;   // sharded runs: n_members <= 2^27
constexpr int MAX_SHARDS = 16;
struct PeerCounts { uint32_t v[MAX_SHARDS]; };   // received records per peer, passed to kernels by value
// How a shard's kernels find what its peers published (passed by value).  direct = 0: the embedder's exchange has copied it
// into this shard's receive buffers (r_recv, q_recv, the replicas), the counts come by value (swimsim_shard_phase*).
// direct = 1 (swimsim_cluster_step: every shard of the cluster is a handle of this process): the kernels read the peers'
// SEND buffers where they lie -- same device, or a peer device over xGMI -- and the counts from the peers' own words:
// nothing is copied but the replicas, nothing comes back to the host.
struct PeerView {
  uint32_t direct;
  const uint4* r[MAX_SHARDS];                 // peer p's round-1 segment (dictionary + lists)
  const uint32_t* rn[MAX_SHARDS];             //   its number of lists (g + G_XLINES)
  const uint2* q[MAX_SHARDS];                 // peer p's round-2 segment for ME
  const uint32_t* qn[MAX_SHARDS];             //   its record count (send_cnt[1][me])
  const unsigned long long* mask[MAX_SHARDS]; // peer p's replica of the queue masks (its own slice is the fresh one)
  const uint8_t* qb[MAX_SHARDS];              //   ... of the queue bytes
  const uint2* st[MAX_SHARDS];                // settling: peer p's list for me, its length (g + G_SETTLE_SEND)
  const uint32_t* stn[MAX_SHARDS];
};
// the same for exchange round 0 (state pulls): peer p's records for me and their number (g + G_JSEND + me)
struct JoinView { uint32_t direct; const uint4* jl[MAX_SHARDS]; const uint32_t* jn[MAX_SHARDS]; };
struct Offsets { uint32_t o[16]; };              // robust scheme: this period's rotation per probe index (0 = none)

constexpr uint32_t SRC_FOREIGN = 1u << 30;      // explicit-record source word: index into fl, not a member

__device__ inline bool is_local(const DevState& s, uint32_t g) { return g - s.lo < s.N; }
__device__ inline uint32_t owner_of(const DevState& s, uint32_t g) { return g / s.N; }   // equal-sized shards

// View cell of local member li in row `slot`.  Rows are tiled over the members: all rows of VTILE consecutive
// members are contiguous ([tile][slot][member in tile]), so a block's accesses to the ~30 rows in circulation
// stay inside one few-MB region (address translation locality) while each row segment is still one contiguous
// run.  SWIM_VTILE = 0 gives plain row-major [slot][member] (measurement knob).
#ifndef SWIM_VTILE
#define SWIM_VTILE 256
#endif
constexpr uint32_t VTILE = SWIM_VTILE;
__host__ __device__ inline size_t vidx_of(uint32_t N, uint32_t R_phys, uint32_t li, uint32_t slot) {
  if (VTILE == 0) return (size_t)slot * N + li;
  return ((size_t)(li / (VTILE ? VTILE : 1u)) * R_phys + slot) * (VTILE ? VTILE : 1u) + (li % (VTILE ? VTILE : 1u));
}
__device__ inline size_t vidx(const DevState& s, uint32_t li, uint32_t slot) { return vidx_of(s.N, s.R_phys, li, slot); }
// ---- view cells: every access goes through these (two layouts, one compile-time switch) -------------------------
// SWIM_VSPLIT = 0: one 8-byte cell {key, lastChange + 1}.  SWIM_VSPLIT = 1: two planes (DevState::Vk / Vs).  The hot
// path (merge_kernel's state rule, the probe's "Alive in my view") sees a cell as VCell {key, tag}: tag = lastChange + 1 in
// full (unsplit) or its low byte (split).  What the rule needs of lastChange, and how the low byte serves it:
//   * "did this entry change in THIS tick already" (CHANGES counts an entry once per tick): tag != (t + 1) & 0xFF says no at
//     once; on a match the full word decides (one more load for a repeated change, or one entry in 256 that last changed a
//     multiple of 256 ticks ago);
//   * a due deadline: the cell of row t mod S names an entry that is Suspect since a tick = t (mod S); for a member that
//     was up all along that tick lies in (t - 256, t] whenever S <= 255, so the low byte names it exactly; S > 255 (a
//     configuration knob nobody uses at these sizes: 3 log2 N = 60 at a million members) loads the full word.
#ifndef SWIM_VSPLIT
#define SWIM_VSPLIT 0
#endif
struct VCell { uint32_t key, tag; };
__device__ inline VCell v_hot(const DevState& s, size_t ix) {
#if SWIM_VSPLIT
  const uint32_t w = s.Vk[ix];
  return VCell{w >> 8, w & 0xFFu};
#else
  const uint2 e = s.V[ix];
  return VCell{e.x, e.y};
#endif
}
__device__ inline uint32_t v_key(const DevState& s, size_t ix) {
#if SWIM_VSPLIT
  return s.Vk[ix] >> 8;
#else
  return s.V[ix].x;
#endif
}
// the full cell {key, lastChange + 1} (cold paths)
__device__ inline uint2 v_full(const DevState& s, size_t ix) {
#if SWIM_VSPLIT
  return make_uint2(s.Vk[ix] >> 8, s.Vs[ix]);
#else
  return s.V[ix];
#endif
}
// -DSWIM_NT_STORES=1 (measurement knob, round 5): merge_kernel's stores -- view cells, queue lines, pk, deadline cells -- as
// non-temporal stores (streamed past the L2: less dirty data for the kernel boundary behind it to write back)
#ifndef SWIM_NT_STORES
#define SWIM_NT_STORES 0
#endif
typedef uint32_t swim_u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t swim_u32x4 __attribute__((ext_vector_type(4)));
__device__ inline void st_u32x2(uint2* p, uint2 v) {
#if SWIM_NT_STORES && defined(__HIP_DEVICE_COMPILE__)
  swim_u32x2 w; w.x = v.x; w.y = v.y;
  __builtin_nontemporal_store(w, reinterpret_cast<swim_u32x2*>(p));
#else
  *p = v;
#endif
}
__device__ inline void st_u32x4(uint4* p, uint4 v) {
#if SWIM_NT_STORES && defined(__HIP_DEVICE_COMPILE__)
  swim_u32x4 w; w.x = v.x; w.y = v.y; w.z = v.z; w.w = v.w;
  __builtin_nontemporal_store(w, reinterpret_cast<swim_u32x4*>(p));
#else
  *p = v;
#endif
}
__device__ inline void v_put(const DevState& s, size_t ix, uint32_t key, uint32_t since1) {
#if SWIM_VSPLIT
  s.Vk[ix] = (key << 8) | (since1 & 0xFFu);
  s.Vs[ix] = since1;
#else
  st_u32x2(&s.V[ix], make_uint2(key, since1));
#endif
}
// has the entry changed in tick t already?  (c = v_hot of the same cell, loaded before)
__device__ inline bool v_changed_in(const DevState& s, size_t ix, const VCell& c, uint32_t t) {
#if SWIM_VSPLIT
  return c.tag == ((t + 1u) & 0xFFu) && s.Vs[ix] == t + 1u;
#else
  (void)s; (void)ix;
  return c.tag == t + 1u;
#endif
}
// a full cell as the hot path sees it
__device__ inline VCell v_cell_of(uint2 e) {
#if SWIM_VSPLIT
  return VCell{e.x, e.y & 0xFFu};
#else
  return VCell{e.x, e.y};
#endif
}
// several writers raise one cell's key (push_kernel: a host with several pullers): returns the key before; the caller that
// raised it stamps lastChange (v_stamp returns the stamp before: the first to stamp counts the change)
__device__ inline uint32_t v_raise_key(const DevState& s, size_t ix, uint32_t km, uint32_t t) {
#if SWIM_VSPLIT
  uint32_t* w = &s.Vk[ix];
  uint32_t cur = atomicOr(w, 0u), old;
  do {
    old = cur;
    if ((old >> 8) >= km) break;
    cur = atomicCAS(w, old, (km << 8) | ((t + 1u) & 0xFFu));
  } while (cur != old);
  return old >> 8;
#else
  (void)t;
  return atomicMax(&reinterpret_cast<uint32_t*>(&s.V[ix])[0], km);
#endif
}
__device__ inline uint32_t v_stamp(const DevState& s, size_t ix, uint32_t t) {
#if SWIM_VSPLIT
  return atomicMax(&s.Vs[ix], t + 1u);
#else
  return atomicMax(&reinterpret_cast<uint32_t*>(&s.V[ix])[1], t + 1u);
#endif
}
// lastChange + 1 of a cell a deadline names (see above)
__device__ inline uint32_t v_since1(const DevState& s, size_t ix, const VCell& c, uint32_t t) {
#if SWIM_VSPLIT
  if (s.S > 255u) return s.Vs[ix];
  return (t + 1u) - (((t + 1u) - c.tag) & 0xFFu);
#else
  (void)s; (void)ix; (void)t;
  return c.tag;
#endif
}
__device__ inline size_t ridx(const DevState& s, uint32_t li, uint32_t pos) {
  return (size_t)pos * s.N + li;
}
__device__ inline const uint4* line_ptr(const DevState& s, uint32_t buf, uint32_t li) {
  return reinterpret_cast<const uint4*>(s.pb + ((size_t)buf * s.N + li) * PB_SLOTS);
}

__device__ inline bool lost(const DevState& s, uint32_t tk, uint32_t purpose, uint32_t src, uint32_t dst,
                            uint32_t idx) {
  if (!s.loss_thr) return false;
  return hash_mk(mix32(tk ^ src), (purpose << 24) | idx, dst) < s.loss_thr;
}

// ---- piggyback line entries -------------------------------------------------------------------
// lo = rumour slot (16) | rumour id (16);  hi = key = inc<<2|state (24) | tx_left (8).  A line holds its
// valid entries first, SORTED by priority (tx desc, subject asc): ageing subtracts the same amount from
// every entry, so the order survives and the owner never needs the subjects of old entries.
__host__ __device__ inline uint32_t pe_slot(uint32_t lo) { return lo & 0xFFFFu; }
__host__ __device__ inline uint32_t pe_rid(uint32_t lo) { return lo >> 16; }
__host__ __device__ inline uint32_t pe_key(uint32_t hi) { return hi & 0xFFFFFFu; }
__host__ __device__ inline uint32_t pe_tx(uint32_t hi) { return hi >> 24; }
__host__ __device__ inline uint32_t pe_lo(uint32_t slot, uint32_t rid) { return slot | (rid << 16); }
__host__ __device__ inline uint32_t pe_hi(uint32_t key, uint32_t tx) { return key | (tx << 24); }

// ---- rumour ids, queue masks and the known-ring --------------------------------------------------
// Every distinct rumour (slot, key) gets a 16-bit id from a global bump counter (G_NRUM) the first time
// any member creates it (rum[id] = {slot, key}), so ids are handed out in time order and the rumours in
// flight at one moment occupy a short id range (measured at 1 M members, one crash per tick: none older
// than 48 ids, profiles/).  H = the counter at the start of the tick (G_HEAD; G_PREV = the tick before).
//
// Transport.  A member's queue is published twice: as the 64-B line and as a 64-bit MASK, bit (rid & 63)
// for every entry with rid in [H-48, H+16) at build time.  "dst merges src's queue" is then ONE
// atomicOr of 8 bytes into inmask[dst] (Pings) or an 8-byte gather from the 8-MB mask table (Acks)
// instead of a 64-B gather from a 64-MB one: the fabric request rate, not bytes, bounds the tick.  A
// receiver decodes bit p as the only id in [H-64, H) with that position, which is exact as long as at
// most 16 ids were allocated since the masks were built (G_HEAD - G_PREV <= MASK_SLACK); otherwise,
// and for queues holding an entry the mask cannot express (MI_OOW), the delivery also travels as an
// explicit record and the receiver reads the source's 64-B line.
//
// Known-ring.  A member keeps a 64-bit ring indexed the same way; only ids in [H-64, H) may be tested
// or set.  Every up member drops, at every tick, the positions of the ids allocated during the
// previous tick ([G_PREV, G_HEAD): the same positions for everybody), so a ring read at tick t is valid
// once those positions are masked off -- by its owner AND by a pinger, which pushes only the bits its
// target does not know yet (atomics run at a fixed ~20-27 G/s on this chip whatever the table size, so
// a skipped push is the cheapest one).  A set bit means "my view entry already dominates this rumour":
// new = incoming & ~known is the whole per-delivery filter.  A clear bit (or an id outside the window)
// only means "look it up".  Ids, masks and the ring are internal: never observable.
#ifndef SWIM_MASK_WIN        // compile-time knobs so that tests can force the fallback paths
#define SWIM_MASK_WIN 48
#define SWIM_MASK_SLACK 16
#endif
#ifndef SWIM_RID_BITS        // width of a rumour id (tests shrink it so that the id counter wraps every few ticks)
#define SWIM_RID_BITS 16
#endif
constexpr uint32_t KN_BITS = 64, MASK_WIN = SWIM_MASK_WIN, MASK_SLACK = SWIM_MASK_SLACK, RID_BITS = SWIM_RID_BITS,
                   RID_MASK = (1u << RID_BITS) - 1u, RID_FAR = 1u << (RID_BITS - 1), RID_NEAR = 1u << (RID_BITS - 2),
                   RID_PARKED = RID_MASK;   // "no id": never handed out, never in a window (an entry too old for the ring)
// the wide known-ring of the explicit-record path (kw, below): ids stay in queue lines for KW_BITS ids behind the head
constexpr uint32_t KW_BITS = RID_NEAR < 256u ? RID_NEAR : 256u;
static_assert(MASK_WIN + MASK_SLACK <= KN_BITS, "mask positions must be unambiguous");
static_assert(RID_BITS <= 16 && KW_BITS + RID_NEAR < RID_FAR + 1 && KN_BITS <= KW_BITS && KW_BITS <= RID_NEAR && (KW_BITS & 63u) == 0u,
              "rumour ids: window < near range < parking distance");
constexpr int RT_WAYS = 8;
constexpr unsigned long long RT_READY = 1ull << 16;
// bits 17..31 of a way: the upper bits of the id counter when the id was handed out, so that the age of a cached id
// is known beyond one turn of the id space (an id exactly 2^RID_BITS allocations old reads as brand-new otherwise)
constexpr uint32_t RT_GEN_SHIFT = 17, RT_GEN_MASK = 0x7FFFu, RT_SPAN_MASK = (1u << (RID_BITS + 15)) - 1u;

// ring / mask position arithmetic (H = head of the tick)
__device__ inline bool rid_in_ring(uint32_t rid, uint32_t H) { return rid != RID_PARKED && ((H - 1u - rid) & RID_MASK) < KN_BITS; }
__device__ inline unsigned long long rid_bit(uint32_t rid) { return 1ull << (rid & 63u); }
// the id in [H-64, H) that owns position p
__device__ inline uint32_t rid_at(uint32_t p, uint32_t H) { return (H - 1u) - ((H - 1u - p) & 63u); }
// ring positions of the ids in [prev, head): what every reader of a ring must disregard this tick
__device__ inline unsigned long long stale_positions(uint32_t prev, uint32_t head) {
  const uint32_t lag = head - prev;
  if (lag >= KN_BITS) return ~0ull;
  if (!lag) return 0ull;
  const unsigned long long run = (1ull << lag) - 1ull;
  const uint32_t sh = prev & 63u;
  return (run << sh) | (sh ? (run >> (64u - sh)) : 0ull);
}
// ---- the wide known-ring (explicit-record path only) -------------------------------------------
// Under message loss queue lines travel as explicit records and a rumour is received ~50 times in its life; the
// 64-position ring forgets it after 64 newer ids (4-6 ticks at 1 % loss and a million members), after which every
// reception costs a scattered view-cell gather -- 48 per member-tick, the bound of that regime.  kw[member] keeps
// "my view dominates the rumour with id r" for the last KW_BITS ids (position r mod KW_BITS), kw_head[member] the
// head it was written at: positions of the ids [kw_head, H) are forgotten when it is read (they stood for ids that
// have left the window).  Loaded and stored only by members that have explicit records this tick.
// four scalar words, not an array: an array member that is indexed inside helper loops stays in memory (the compiler kept
// `w[4]` in scratch -- 40 bytes per lane and 53 scratch instructions in merge_kernel -- or moved it to LDS)
struct Ring256 { unsigned long long w0, w1, w2, w3; };
static_assert(KW_BITS == 256u || KW_BITS == 128u || KW_BITS == 64u, "the wide ring is one, two or four words");
__device__ inline bool rid_in_wide(uint32_t rid, uint32_t H) { return rid != RID_PARKED && ((H - 1u - rid) & RID_MASK) < KW_BITS; }
__device__ inline unsigned long long low_bits(int n) { return n <= 0 ? 0ull : n >= 64 ? ~0ull : (1ull << n) - 1ull; }
// the positions of word k that the circular range [a, a + len) of a KW_BITS-position ring covers
__device__ inline unsigned long long r256_range_word(uint32_t a, uint32_t len, int k) {
  if (len >= KW_BITS) return ~0ull;
  const int off = (int)(((uint32_t)(64 * k) - a) & (KW_BITS - 1u));   // distance of the word's first position from a
  unsigned long long m = low_bits((int)len - off);
  if (off > (int)KW_BITS - 64) m |= low_bits((int)len + (int)KW_BITS - off) & ~low_bits((int)KW_BITS - off);
  return m;
}
// forget the circular position range [a, a + len) of a KW_BITS-position ring
__device__ inline void r256_forget(Ring256& r, uint32_t a, uint32_t len) {
  a &= KW_BITS - 1u;
  r.w0 &= ~r256_range_word(a, len, 0);
  if (KW_BITS > 64u) r.w1 &= ~r256_range_word(a, len, 1);
  if (KW_BITS > 128u) { r.w2 &= ~r256_range_word(a, len, 2); r.w3 &= ~r256_range_word(a, len, 3); }
}
__device__ inline bool r256_test(const Ring256& r, uint32_t rid) {
  const uint32_t q = rid & (KW_BITS - 1u), k = q >> 6;
  const unsigned long long w = k == 0u ? r.w0 : k == 1u ? r.w1 : k == 2u ? r.w2 : r.w3;
  return (w >> (q & 63u)) & 1ull;
}
__device__ inline void r256_or(Ring256& r, uint32_t word, unsigned long long m) {
  r.w0 |= word == 0u ? m : 0ull; r.w1 |= word == 1u ? m : 0ull; r.w2 |= word == 2u ? m : 0ull; r.w3 |= word == 3u ? m : 0ull;
}
__device__ inline void r256_set(Ring256& r, uint32_t rid) { r256_or(r, (rid & (KW_BITS - 1u)) >> 6, 1ull << (rid & 63u)); }

// can an entry with this id be expressed in a mask built at head H?
__device__ inline bool rid_maskable(uint32_t rid, uint32_t H) {
  return rid != RID_PARKED && ((rid - (H - MASK_WIN)) & RID_MASK) < MASK_WIN + MASK_SLACK;
}

// minfo fields
constexpr uint32_t MI_SLOT = 0xFFFFu, MI_PBN_SHIFT = 16, MI_PBN = 0xFu << 16, MI_BUF = 1u << 20,
                   MI_UP = 1u << 21, MI_OOW = 1u << 22, MI_PB = MI_PBN | MI_BUF | MI_OOW,
                   MI_BASE_SHIFT = 23, MI_BASE = 3u << MI_BASE_SHIFT;
constexpr uint32_t MB_UP = 1u, MB_PBN_SHIFT = 1, MB_OOW = 1u << 5, MB_ROW = 1u << 6, MB_BASE_NA = 1u << 7;
__host__ __device__ inline uint32_t mb_of(uint32_t mi) {
  return ((mi & MI_UP) ? MB_UP : 0u) | (((mi >> MI_PBN_SHIFT) & 0xFu) << MB_PBN_SHIFT) | ((mi & MI_OOW) ? MB_OOW : 0u) |
         ((mi & MI_SLOT) ? MB_ROW : 0u) | ((mi & MI_BASE) ? MB_BASE_NA : 0u);
}
__device__ inline void set_mi(const DevState& s, uint32_t g, uint32_t v) { s.minfo[g] = v; s.mb[g] = (uint8_t)mb_of(v); }
// the member now has a view row (minfo was updated with an atomic by whoever allocated it)
__device__ inline void mb_set_row(const DevState& s, uint32_t g) {
  atomicOr(reinterpret_cast<uint32_t*>(s.mb) + (g >> 2), MB_ROW << (8u * (g & 3u)));
}
// minfo of a probe target / proxy as far as the prober needs it: from the byte table, the full word only for
// members with a view row or an unmaskable queue and on ticks that travel as explicit records (the line
// buffer bit is needed then)
// ... from the member's byte b = mb[c], gathered by the caller (probe_kernel asks for the first draws' bytes with its own word)
__device__ inline uint32_t probe_mi_byte(const DevState& s, uint32_t c, uint32_t b, bool use_mask) {
  if ((b & (MB_ROW | MB_OOW)) || !use_mask) return s.minfo[c];
  return ((b & MB_UP) ? MI_UP : 0u) | (((b >> MB_PBN_SHIFT) & 0xFu) << MI_PBN_SHIFT) |
         ((b & MB_BASE_NA) ? ((uint32_t)ST_DEAD << MI_BASE_SHIFT) : 0u);
}
__device__ inline uint32_t probe_mb(const DevState& s, uint32_t c) {
  return ABL(ABL_MB_GATHER) ? (MB_UP | (8u << MB_PBN_SHIFT)) : (uint32_t)s.mb[c];
}
__device__ inline uint32_t probe_mi(const DevState& s, uint32_t c, bool use_mask) {
#ifdef SWIM_NO_MB            // measurement knob: always gather the full word
  return s.minfo[c];
#endif
  return probe_mi_byte(s, c, probe_mb(s, c), use_mask);
}
__device__ inline uint32_t mi_pbn(uint32_t mi) { return (mi >> MI_PBN_SHIFT) & 0xFu; }
__device__ inline uint32_t mi_buf(uint32_t mi) { return (mi >> 20) & 1u; }
__device__ inline bool mi_up(uint32_t mi) { return (mi & MI_UP) != 0; }
// source word for "merge this member's current piggyback line": id | buffer<<31
__device__ inline uint32_t mi_src(uint32_t id, uint32_t mi) { return id | (mi_buf(mi) << 31); }

// `isAlive` on local member li's view of c (src/Core.hs:33-34, 72-74); mc = minfo[c]
__device__ inline bool view_alive(const DevState& s, uint32_t li, uint32_t mc) {
  const uint32_t sl = mc & MI_SLOT;
  if (sl == 0 || sl == MI_SLOT) return ((mc >> MI_BASE_SHIFT) & 3u) == ST_ALIVE;   // no row: the settled base (Alive@0 at first)
  const uint32_t k = v_key(s, vidx(s, li, sl - 1));
  return ((k ? k : s.slot_base[sl - 1]) & 3u) == ST_ALIVE;
}

// ---- suspicion deadlines (trow) ---------------------------------------------------------------
// One 16-byte cell per (deadline mod S, member): halfwords 0..6 = slot+1 of up to 7 deadlines, packed from 0;
// halfword 7 = link: 0 none, TR_FULL = "look at every view row" (pool exhausted: exact, slow), otherwise
// TR_LINK | high bits: the chain continues in overflow cell idx = (hw7 & 0x7FFF) << 16 | hw6 of the row's pool
// (tovf; a member that accepts more than 7 suspicions in one tick -- heavy message loss) and the cell holds
// 6 deadlines (a 15-bit index ran out at 1 % loss and a million members: 48 000 members per tick fell back to
// the full scan, profiles/r02y_batching_variants_and_lossy_events.txt).  Pools alternate by cycle parity (t / S) & 1: tick t consumes
// the chains written at t - S from pool parity^1 while it writes the chains for t + S into pool parity.
constexpr uint32_t TR_PAY = 7, TR_FULL = 0xFFFFu, TR_LINK = 0x8000u;
struct TimerCell { unsigned long long lo, hi; uint32_t n; };
__device__ inline void tc_clear(TimerCell& c) { c.lo = 0; c.hi = 0; c.n = 0; }
__device__ inline void tc_set(TimerCell& c, uint32_t pos, uint32_t v) {       // pos 0..7
  if (pos < 4u) c.lo |= (unsigned long long)v << (16u * pos);
  else c.hi |= (unsigned long long)v << (16u * (pos - 4u));
}
// without a pool (fixtures, cells rebuilt after a downtime): the 8th deadline turns the cell into "look everywhere"
__device__ inline bool tc_linked(uint32_t w3) { const uint32_t l = w3 >> 16; return (l & TR_LINK) != 0u && l != TR_FULL; }
__device__ inline uint32_t tc_link_idx(uint32_t w3) { return (((w3 >> 16) & (TR_LINK - 1u)) << 16) | (w3 & 0xFFFFu); }
__device__ inline uint32_t tc_cap(const TimerCell& c) { return tc_linked((uint32_t)(c.hi >> 32)) ? TR_PAY - 1u : TR_PAY; }
__device__ inline void tc_set_link(TimerCell& c, uint32_t idx) {     // on a cleared cell
  c.hi = ((unsigned long long)(TR_LINK | (idx >> 16)) << 48) | ((unsigned long long)(idx & 0xFFFFu) << 32);
}
__device__ inline void tc_put_simple(TimerCell& c, uint32_t slot1) {
  if (c.n < tc_cap(c)) tc_set(c, c.n, slot1);
  else c.hi |= (unsigned long long)TR_FULL << 48;
  c.n++;
}
__device__ inline uint32_t tc_get(const uint4& v, uint32_t k) {
  if (k == TR_PAY - 1u && tc_linked(v.w)) return 0u;                  // that halfword is part of the link
  const uint32_t w = k < 2u ? v.x : k < 4u ? v.y : k < 6u ? v.z : v.w;
  return (k & 1u) ? (w >> 16) : (w & 0xFFFFu);
}
__device__ inline uint4 tc_pack(const TimerCell& c) {
  return make_uint4((uint32_t)c.lo, (uint32_t)(c.lo >> 32), (uint32_t)c.hi, (uint32_t)(c.hi >> 32));
}

// (forward: the overload without first draws gathered ahead is below select_members)
// the first draw of each of the n picks of kRandomMembers(.., P_SELECT) and its byte of the mb table: pure functions of (tick, member),
// so probe_kernel asks for them in the SAME round of loads as the member's own word (round 6: they were a round trip of their own)
template <int MAXN>
__device__ inline void select_first_draws(const DevState& s, uint32_t mk, uint32_t n, uint32_t (&c)[MAXN], uint32_t (&b)[MAXN]) {
#pragma unroll
  for (int p = 0; p < MAXN; ++p) {
    c[p] = 0; b[p] = 0;
    if ((uint32_t)p < n) {
      c[p] = __umulhi(hash_mk(mk, ((uint32_t)P_SELECT << 24) | ((uint32_t)p << 8), 0), s.NT);
      b[p] = probe_mb(s, c[p]);
    }
  }
}

// kRandomMembers (src/Core.hs:69-74) + shuffle (src/Util.hs:37-42) as n draws without
// replacement: rejection sampling on the counter RNG, then a cyclic scan so that "fewer than
// n candidates => all of them" holds exactly (test/Spec.hs:117-128).  Self never eligible (D15).
template <int MAXN>
__device__ inline uint32_t select_members(const DevState& s, uint32_t mk, uint32_t i, uint32_t n,
                                          uint32_t purpose, uint32_t hi_idx, const uint32_t* excl,
                                          uint32_t nexcl, uint32_t (&out)[MAXN],
                                          uint32_t (&info)[MAXN], bool use_mask, bool* all_first,
                                          bool pre, const uint32_t (&pre_c)[MAXN], const uint32_t (&pre_b)[MAXN]) {
  // pre: pre_c / pre_b hold the first draw of every index and its byte of `mb`, gathered by the caller ahead of time (select_first_draws)
  uint32_t np = 0;
  bool first_only = true;          // every pick so far was the first draw of its index
  const uint32_t N = s.NT;
  // Issue the first-attempt gathers of all picks together (independent loads); eligibility is
  // then decided pick by pick in order, exactly as the sequential definition does.
  uint32_t c0[MAXN <= 16 ? MAXN : 1], m0[MAXN <= 16 ? MAXN : 1];
  if (MAXN <= 16) {
#pragma unroll
    for (int p = 0; p < (MAXN <= 16 ? MAXN : 1); ++p) {
      c0[p] = 0; m0[p] = 0;
      if ((uint32_t)p < n) {
        const uint32_t base = (purpose << 24) | (purpose == P_SELECT ? ((uint32_t)p << 8) : ((hi_idx << 16) | ((uint32_t)p << 8)));
        c0[p] = pre ? pre_c[p] : __umulhi(hash_mk(mk, base, 0), N);
#ifdef SWIM_NO_MB
        m0[p] = probe_mi(s, c0[p], use_mask);
#else
        m0[p] = pre ? probe_mi_byte(s, c0[p], pre_b[p], use_mask) : probe_mi(s, c0[p], use_mask);
#endif
      }
    }
  }
  for (uint32_t p = 0; p < (uint32_t)MAXN; ++p) {
    if (p >= n) break;
    uint32_t c = 0, mc = 0;
    bool found = false;
    const uint32_t base = (purpose << 24) | (purpose == P_SELECT ? (p << 8) : ((hi_idx << 16) | (p << 8)));
    auto eligible = [&](uint32_t cand, bool have, uint32_t mhave) -> bool {
      if (cand == i) return false;
      for (uint32_t e = 0; e < nexcl; ++e) if (excl[e] == cand) return false;
      bool dup = false;
      for (int e = 0; e < MAXN; ++e) dup |= ((uint32_t)e < np) && (out[e] == cand);
      if (dup) return false;
      mc = have ? mhave : probe_mi(s, cand, use_mask);
      return view_alive(s, i - s.lo, mc);
    };
    for (uint32_t a = 0; a < SEL_ATTEMPTS; ++a) {
      bool have = false; uint32_t mh = 0;
      if (MAXN <= 16 && a == 0) {
        have = true;
        for (int e = 0; e < (MAXN <= 16 ? MAXN : 1); ++e) if ((uint32_t)e == p) { c = c0[e]; mh = m0[e]; }
      } else {
        c = __umulhi(hash_mk(mk, base | a, 0), N);
      }
      if (eligible(c, have, mh)) { found = true; first_only &= a == 0u; break; }
    }
    if (!found) {
      first_only = false;
      uint32_t cs = (c + 1 == N) ? 0 : c + 1;
      for (uint32_t d = 0; d < N; ++d) {
        c = cs + d; if (c >= N) c -= N;
        if (eligible(c, false, 0)) { found = true; break; }
      }
    }
    if (!found) break;
    for (int e = 0; e < MAXN; ++e) if ((uint32_t)e == np) { out[e] = c; info[e] = mc; }
    ++np;
  }
  if (all_first) *all_first = first_only && np == n;
  return np;
}
template <int MAXN>
__device__ inline uint32_t select_members(const DevState& s, uint32_t mk, uint32_t i, uint32_t n,
                                          uint32_t purpose, uint32_t hi_idx, const uint32_t* excl,
                                          uint32_t nexcl, uint32_t (&out)[MAXN],
                                          uint32_t (&info)[MAXN], bool use_mask = false, bool* all_first = nullptr) {
  uint32_t none[MAXN];
#pragma unroll
  for (int p = 0; p < MAXN; ++p) none[p] = 0;
  return select_members<MAXN>(s, mk, i, n, purpose, hi_idx, excl, nexcl, out, info, use_mask, all_first, false, none, none);
}

}  // namespace swim
