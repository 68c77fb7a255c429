/*
 * swimsim.h -- C ABI of the MI355X-native SWIM tick simulator (libswimsim.so).
 *
 * This is the drop-in boundary for ONE path of jpfuentes2/swim: the gossip /
 * failure-detection round (probe -> ack -> k-indirect ping-req -> suspicion
 * timer -> piggyback dissemination), stepped as a bulk-synchronous tick over
 * N simulated members.  The reference has no FFI of its own:
 * its seams are three conduit-typed Haskell functions.  Each entry point below
 * names the reference interface it replaces (paths relative to the reference
 * checkout).  The Haskell-side binding a maintainer would add is shown in
 * INTEGRATION.md and shipped as source in haskell/Swim/Sim.hs.
 *
 * Conventions
 *   - plain C, fixed-width integers, no C++ and no torch types in signatures;
 *   - every call returns SWIMSIM_OK (0) or a negative swimsim_status; the text
 *     is available from swimsim_last_error();  no C++ exception crosses the ABI
 *     (reference style: `configure :: IO (Either Error Store)`, src/Util.hs:103);
 *   - the caller allocates every output buffer and passes its capacity;
 *   - one handle = one logical thread of control (guard with an MVar on the
 *     Haskell side); different handles are independent;
 *   - member names of the reference (`memberName :: String`, src/Types.hs:62)
 *     are rendered as "m<id>", id in [0, n_members).
 *
 * The library is the HIP/gfx950 implementation only.  There is NO CPU fallback:
 * creating a handle without a usable GPU fails with SWIMSIM_ERR_DEVICE.
 */
#ifndef SWIMSIM_H
#define SWIMSIM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 7: the embedder-facing exchange of dense shards (round 5: SWIMSIM_PREC_BYTES 16 -> 8, kind 0 = one segment for every peer, kind 2
 * gone, the kind-5/6 gathers mandatory in round 1 of swimsim_shard_step) + swimsim_note_outside_rumor (round 6) -- a host built against
 * the header of version 6 would copy the wrong strides, so it is refused at swimsim_create instead; 6: config fields
 * strict_reference_rules, push_pull (round 4); 5: view_cap, counter EVICTED, swimsim_cluster_step, the wire codec's
 * bare form (round 4); 4: pull_ticks, swimsim_inject_rumor, the bridge (round 3).  A handle is refused unless struct_size and
 * abi_version match the library's. */
#define SWIMSIM_ABI_VERSION 7u

/* ---- status codes ------------------------------------------------------ */
typedef enum swimsim_status {
  SWIMSIM_OK = 0,
  SWIMSIM_ERR_INVALID = -1,   /* bad argument / config                          */
  SWIMSIM_ERR_DEVICE = -2,    /* no GPU, HIP error                              */
  SWIMSIM_ERR_NOMEM = -3,     /* host or device allocation failed               */
  SWIMSIM_ERR_CAPACITY = -4,  /* a bounded table overflowed (max_subjects,
                                 inbox overflow list, incarnation bits, shard
                                 exchange buffers); the handle is poisoned
                                 afterwards                                     */
  SWIMSIM_ERR_STATE = -5,     /* call not valid in this state (poisoned handle) */
  SWIMSIM_ERR_BUFFER = -6     /* caller buffer too small (n_out has the need)   */
} swimsim_status;

/* ---- Liveness (src/Types.hs:76 `data Liveness = IsAliveC | IsSuspectC | IsDeadC`) */
enum { SWIMSIM_ALIVE = 0, SWIMSIM_SUSPECT = 1, SWIMSIM_DEAD = 2 };

/* Causes of a membership event (which rule of the reference produced it). */
enum {
  SWIMSIM_CAUSE_PROBE = 0,   /* own probe + k indirect probes failed: src/Core.hs:253-254 */
  SWIMSIM_CAUSE_TIMER = 1,   /* suspicion timeout -> Dead: the FIXME at src/Core.hs:141    */
  SWIMSIM_CAUSE_GOSSIP = 2,  /* piggybacked Suspect/Alive/Dead accepted: src/Core.hs:110-117 */
  SWIMSIM_CAUSE_REFUTE = 3,  /* rumour about self refuted: src/Core.hs:155-166          */
  SWIMSIM_CAUSE_JOIN = 4     /* member came (back) up and announced Alive              */
};
#define SWIMSIM_EVMASK_ALL 0x1Fu
#define SWIMSIM_EVMASK_DEFAULT \
  ((1u << SWIMSIM_CAUSE_PROBE) | (1u << SWIMSIM_CAUSE_REFUTE) | (1u << SWIMSIM_CAUSE_JOIN))

/* ---- configuration ------------------------------------------------------
 * First block = the reference's `Config` (src/Types.hs:46-51) as far as the hot
 * path reads it (numToGossip, gossipInterval: src/Core.hs:237,239,249,258).
 * bindHost / joinHosts / udpBufferSize are dead fields in the reference and live
 * only in the host-language mirror (swim_amd.types.Config, haskell/Swim/Sim.hs).
 * Second block = simulator-only knobs ("SimConfig").
 * A field left 0 takes the documented default.
 */
typedef struct swimsim_config {
  uint32_t struct_size;        /* = sizeof(swimsim_config_t)                          */
  uint32_t abi_version;        /* = SWIMSIM_ABI_VERSION                               */
  /* -- reference Config -- */
  int32_t  num_to_gossip;      /* numToGossip: probes per period AND proxies per failed
                                  probe (one field for both, src/Core.hs:239,249)     */
  int64_t  gossip_interval_us; /* gossipInterval; 1 tick == 1 interval (D1)           */
  /* -- simulator -- */
  uint32_t n_members;          /* N >= 2                                              */
  uint64_t seed;               /* counter-based RNG seed (replaces global StdGen, F7) */
  int32_t  probes_per_tick;    /* 0 -> num_to_gossip (D14)                            */
  int32_t  indirect_k;         /* 0 -> num_to_gossip (D7)                             */
  uint32_t loss_ppm;           /* per-message loss probability, parts per million     */
  uint32_t suspicion_ticks;    /* Suspect -> Dead timeout (D4); 0 -> 3*ceil(log2 N)   */
  uint32_t retransmit_mult;    /* piggyback budget L = mult*ceil(log2(N+1)) (D5); 0->3 */
  uint32_t max_subjects;       /* subjects with a live view column at one time (columns are
                                  reclaimed by settling, gc_ticks); 0 -> sized for the false
                                  suspicions of ~256 periods at loss_ppm (>= 1024, <= N, <= 60000,
                                  within 32 GB of columns)                                  */
  uint32_t gc_ticks;           /* settling horizon G (`removeDeadNodes`, src/Core.hs:65-67, plus
                                  the push-pull anti-entropy the reference leaves commented out,
                                  src/Types.hs:165,177): a subject nobody has changed its mind
                                  about for G periods is reconciled and its view column
                                  reclaimed.  0 -> off, SWIMSIM_GC_AUTO -> the minimum
                                  suspicion_ticks + L + 2; smaller values are refused         */
  uint32_t event_cap;          /* event ring capacity; 0 -> 1<<20                     */
  uint32_t event_mask;         /* bit per SWIMSIM_CAUSE_*; 0 -> SWIMSIM_EVMASK_DEFAULT */
  uint32_t inbox_cap;          /* per-member delivery slots per tick; 0 -> sized from the
                                  expected fan-in (2P + 4PK*P[direct probe fails]); rarer
                                  excess goes through an exact overflow list             */
  int32_t  device;             /* HIP device ordinal                                  */
  uint32_t shard_index;        /* this handle owns members [shard_index*n_members/n_shards,
                                  ...) of the population; n_members is the WHOLE population */
  uint32_t n_shards;           /* 0/1 -> unsharded; <= 16, must divide n_members      */
  uint32_t target_scheme;      /* SWIMSIM_TARGETS_RANDOM (the reference: kRandomMembers,
                                  src/Core.hs:239) or SWIMSIM_TARGETS_ROBUST (the FIXME at
                                  src/Core.hs:232 "move from random to robust scheme")   */
  uint32_t join_pull;          /* 1: a member that comes (back) up pulls the member map of a join host
                                  (`joinHosts`, src/Types.hs:47, src/Util.hs:46; the commented-out
                                  PushPullMsg, src/Types.hs:165,177) -- see below.  0: it keeps the map
                                  it had and learns the rest from gossip                       */
  uint32_t pull_ticks;         /* T > 1: periodic state pull between up members, once per member every T
                                  periods (memberlist's push-pull timer; the commented-out PushPullMsg,
                                  src/Types.hs:165,177) -- see below.  0: off.  1: invalid            */
  uint32_t view_cap;           /* C > 0: BOUNDED member maps -- every member keeps at most C entries that differ
                                  from the default (`Map String Member`, src/Types.hs:55, with a capacity), in
                                  [SWIMSIM_VIEW_CAP_MIN, SWIMSIM_VIEW_CAP_MAX]; see "Bounded member maps" below.
                                  0: unbounded (a view row per subject in circulation, max_subjects)   */
  uint32_t strict_reference_rules; /* 1: the LITERAL suspectOrDeadNode' (src/Core.hs:142-187) instead of the commutative
                                  merge -- see "Strict reference rules" below.  0: the merge   */
  uint32_t push_pull;          /* 1 (needs pull_ticks): the periodic state pull is a PUSH-PULL -- the host merges the puller's
                                  member map in the same exchange (the push half of the commented-out PushPullMsg,
                                  src/Types.hs:165,177) -- see "Periodic state pull" below.  0: pull only               */
} swimsim_config_t;

/* Strict reference rules (strict_reference_rules = 1; DESIGN.md sections 2.9 and 3, D13).  The reference's state rule is not the
 * max-merge: beyond "an older incarnation is ignored" (src/Core.hs:151) it IGNORES a Suspect unless the entry is Alive (livenessCheck
 * IsSuspect, :183) and a Dead when the entry is Dead already (IsDead, :184) -- at ANY incarnation, so a Suspect at a higher
 * incarnation than a Suspect / Dead entry, or a Dead at a higher incarnation than a Dead entry, changes nothing there (the merge takes
 * them).  That rule depends on the order in which a member's proposals of one period are applied, which the reference leaves to its
 * scheduler; here the order is CANONICAL: the member's due suspicion deadlines, then its own failed probes, then the rumours delivered
 * to it sorted by (subject, incarnation<<2|state) ascending, each applied to what the ones before it left.  (Alive rumours follow the
 * merge in both modes: aliveNode is unwritten, src/Core.hs:197-218, D6; rumours about the member itself go to the refutation rule.)
 * Because an ignored rumour may be accepted later -- after a refutation has made the entry Alive again -- no delivery may be filtered
 * as "known already": every delivered queue entry is examined every time (the handle runs the exact record path in every tick), so a
 * tick costs several times the default's.  Sharded clusters (round 6): across shards every queue travels as a list of (subject, key) and the owner
 * of a member applies what was delivered to it in the same canonical order -- the cluster equals the unsharded run.  Settling, the
 * join-time pull and the periodic pull / push-pull combine with it (round 6): they are state transfer, not rumours -- a pulled or pushed
 * entry follows the merge of the pull (the larger entry wins) in both modes, settling reconciles to the largest entry among the members
 * that are up as always.  Not combinable with view_cap (SWIMSIM_ERR_INVALID). */

/* Bounded member maps (view_cap = C > 0) -- what lets heavy message loss run at millions of members
 * per GPU (BASELINE config 5): there nearly every member is a subject of somebody's false suspicion all the time, and a
 * dense view row per subject cannot exist.  A member's map holds at most C exceptions {subject, incarnation<<2|state,
 * lastChange} to the default "Alive at incarnation 0".  The end of a tick for member i, as a function of SETS (so that no
 * processing order can matter):
 *   proposals = its suspicion deadlines that are due (an entry Suspect since t' with t' + suspicion_ticks <= t -> Dead at the
 *               same incarnation), its own probes that ended without an Ack (Suspect at the incarnation it holds), every
 *               rumour delivered to it (rumours about itself go to the refutation rule as always);
 *   for every subject the largest proposal wins against the entry (or the default) by the state rule's order; an entry that
 *               grows is CHANGED: lastChange = t;
 *   if more than C entries exist now, the C with the largest (lastChange, rank) stay, rank = mix32(subject ^ mix32(tk ^ i))
 *               with the tick key tk of DESIGN.md 2.2 -- a keyed permutation of the subject ids, new for every member
 *               and tick, so that no two entries tie and nobody is forgotten first by everybody; the others are EVICTED --
 *               they return to the default (their deadlines with them); counter SWIMSIM_CTR_EVICTED counts the entries of the
 *               START of the tick that leave.  An entry that appeared in this very tick and is evicted at once has no effect
 *               at all (no event, no counter, not gossiped on), and neither has the change of an older entry that leaves;
 *   every entry that changed and stayed counts as a view change (counters, event -- cause of the winning proposal, a due
 *               deadline before a probe before gossip when they propose the same --, digest) and becomes a rumour in the
 *               member's queue with a full retransmission budget, as on an unbounded handle.
 * Everything else of the tick (target selection among the members the map holds Alive -- evicted ones are Alive again --,
 * probes, proxies, loss, queues, refutation, joins) is the unbounded tick's.  Not combinable with gc_ticks, join_pull,
 * pull_ticks, the robust target scheme, swimsim_inject_rumor, swimsim_set_view and swimsim_k_random_members
 * (SWIMSIM_ERR_INVALID).  Sharded clusters of bounded handles (n_shards > 1) use the same phase calls as dense ones with a
 * simpler exchange -- see "sharded clusters" below. */
#define SWIMSIM_VIEW_CAP_MIN 4u
#define SWIMSIM_VIEW_CAP_MAX 256u

/* Join-time state pull (join_pull = 1).  When member m comes up in tick t its join host
 * is the first of the 8 draws mulhi(H(t, m, JOIN<<24 | a, 0), N), a = 0..7, that is not m, was up before this
 * tick and has no scheduled change in this tick (so that the result does not depend on the order in which the
 * tick's changes are applied); none => no pull.  For every subject s != m that has a view row, m's entry becomes
 * max(own entry, the host's entry) -- the host itself counts as Alive at its own incarnation --, with
 * lastChange = t and a suspicion deadline t + suspicion_ticks for a pulled Suspect.  A pulled entry is a view
 * change like any other (counters, digest), but is not gossiped on and raises no event.  On sharded handles the
 * host may live on another shard: its owner sends what it knows in exchange round 0 (swimsim_shard_phase0 below). */

/* Periodic state pull (pull_ticks = T > 1) -- anti-entropy between members that are up, the pull
 * half of memberlist's push-pull.  In tick t, after the tick's scheduled changes and the joiners' pulls, every member i
 * with i mod T == t mod T that is up and has no scheduled change in this tick pulls from the first of the 8 draws
 * mulhi(H(t, i, PULL<<24 | a, 0), N), a = 0..7, that is not i, is up, has no scheduled change in this tick and is not a
 * puller of this tick itself (c mod T != t mod T) -- so nobody reads a member map that is being written, and the result
 * does not depend on any order; none => no pull this period.  The merge is the join-time pull's: for every subject s != i
 * with a view row, i's entry becomes max(own entry, the host's entry), the host counting as Alive at its own incarnation;
 * lastChange = t, a pulled Suspect gets the deadline t + suspicion_ticks; a view change like any other (counters, digest),
 * not gossiped on, no event.  With pull_ticks on, a join host is not one of the tick's periodic pullers either.
 * push_pull = 1 adds the PUSH half (memberlist's pushPull sends the local state and merges the remote one on both sides): after ALL
 * of the tick's pulls, every puller's host merges the puller's member map the same way -- for every subject s != host with a view
 * row, the host's entry becomes max(own entry, the puller's entry), the puller counting as Alive at its own incarnation; lastChange
 * = t, deadline t + suspicion_ticks for a pushed Suspect, a view change like any other, not gossiped on, no event.  A host may have
 * several pullers in one tick: the result is the max over all of them -- no order can matter, because hosts are never pullers (the
 * pulls read maps nobody writes, the pushes write maps nobody reads), and pushing the map AFTER the pull gives the host exactly
 * what the map before the pull would have (max is idempotent).
 * On sharded handles the periodic pull works as the join-time pull does: a puller whose host lives on another shard receives the host's
 * map as kind-4 records in exchange round 0, so a shard with pull_ticks starts EVERY tick with swimsim_shard_phase0 (swimsim_shard_step
 * does).  push_pull on sharded handles (round 6): a puller whose host lives on another shard hands the host's owner its map as records of
 * the same round 0 ({host | 1 << 31, subject, entry}); pairs of one shard push as on one handle. */

#define SWIMSIM_GC_AUTO 0xFFFFFFFFu

/* Settling (gc_ticks = G > 0).  At the end of tick t, every subject s with a
 * view column whose last change in ANY view (or last announcement: refutation, join) is older than
 * t - G is settled: k* = the largest entry about s among the members that are up (the value their
 * views would converge to under push-pull anti-entropy), base(s) := max(base(s), k*), and every
 * member's entry about s returns to the default -- which from then on means base(s), not Alive@0.  A
 * settled Dead subject is thereby removed from every member map (`removeDeadNodes`) and survives only
 * as a population-wide tombstone (never picked, rumours at its incarnation ignored, an Alive at a
 * higher incarnation re-adds it); a settled Alive@i subject stays listed with since_tick = the settling
 * tick.  Nothing is settled while a member that is up still holds it Suspect.  The column is reusable
 * two ticks later.  A subject has a column from the first rumour anybody states about it -- a failed probe's
 * suspicion, a delivered rumour, its OWN join announcement (a member that joins and goes down again before anybody
 * hears of it leaves an empty column, which settles like any other: counter SETTLED, max_subjects).  G >= suspicion_ticks + L + 2 guarantees that no piggyback queue and no pending
 * timer of an up member refers to s any more.  On sharded handles the decision needs every shard's word: a third,
 * small exchange round per tick (swimsim_shard_settle_*, below). */

/* Target schemes for the direct probes of a period.
 * RANDOM: numToGossip members drawn uniformly among those Alive in the prober's view (the reference).
 * ROBUST: the round-robin selection of the SWIM paper (section 4.3) as a population-wide rotation:
 *   in period t probe p of member i goes to (i + o(t,p)) mod N, where the offsets o run through a
 *   pseudo-random permutation of 1..N-1 in rounds of ceil((N-1)/numToGossip) periods (DESIGN.md
 *   section 9).  Every member is probed by exactly numToGossip members per period and every member
 *   probes every other one once per round: detection time is bounded, and since the pingers of a
 *   member are computable its Ping payloads are pulled instead of pushed (no atomics).  Targets that
 *   are not Alive in the prober's view are skipped, proxies stay uniformly random.  On sharded handles the
 *   targets are the same and the payloads are pushed (the pinger may live on another shard). */
enum { SWIMSIM_TARGETS_RANDOM = 0, SWIMSIM_TARGETS_ROBUST = 1 };

typedef struct swimsim swimsim_t; /* opaque; owned by the library */

/* Membership event = the `Broadcast (Suspect|Alive|Dead ...)` gossip a node of the
 * reference would enqueue (src/Types.hs:42-44,135-145; src/Core.hs:119-121,254). */
typedef struct swimsim_event {
  uint64_t tick;
  uint32_t observer;     /* whose view changed (for REFUTE/JOIN == subject)            */
  uint32_t subject;      /* `node`                                                     */
  uint32_t incarnation;  /* `incarnation`                                              */
  uint8_t  state;        /* SWIMSIM_ALIVE / SUSPECT / DEAD                             */
  uint8_t  cause;        /* SWIMSIM_CAUSE_*                                            */
  uint16_t _pad;
} swimsim_event_t;

/* One non-default entry of a member's view = `Member` (src/Types.hs:62-68). */
typedef struct swimsim_view_entry {
  uint32_t subject;      /* memberName = "m<subject>"                                  */
  uint32_t incarnation;  /* memberIncarnation                                          */
  uint32_t since_tick;   /* memberLastChange, in ticks                                 */
  uint8_t  state;        /* memberAlive                                                */
  uint8_t  _pad[3];
} swimsim_view_entry_t;

typedef struct swimsim_rumor {
  uint32_t subject;
  uint32_t incarnation;
  uint8_t  state;
  uint8_t  tx_left;      /* remaining piggyback transmissions                          */
  uint16_t _pad;
} swimsim_rumor_t;

/* Per-member summary: the parts of `Store` (src/Types.hs:53-60) that survive in a
 * tick model: storeIncarnation, plus the piggyback queue the reference leaves as
 * a FIXME (src/Core.hs:136-138). */
typedef struct swimsim_member {
  uint32_t id;
  uint32_t incarnation;       /* storeIncarnation                                      */
  uint8_t  up;                /* ground truth (fault schedule), not protocol state     */
  uint8_t  n_rumors;
  uint16_t n_timers;          /* pending suspicion timers                              */
  swimsim_rumor_t rumors[8];  /* piggyback buffer, order unspecified                   */
} swimsim_member_t;

/* Counter indices for swimsim_counters(). */
enum {
  SWIMSIM_CTR_PINGS = 0,          /* direct Pings sent (src/Core.hs:246)                 */
  SWIMSIM_CTR_DIRECT_FAILED = 1,  /* f: direct probes not acked                          */
  SWIMSIM_CTR_PING_REQS = 2,      /* IndirectPings sent (src/Core.hs:250)                */
  SWIMSIM_CTR_SUSPECTS = 3,       /* probes that ended in Suspect (src/Core.hs:253)      */
  SWIMSIM_CTR_FALSE_SUSPECTS = 4, /* ... of a member that was actually up                */
  SWIMSIM_CTR_PAYLOADS = 5,       /* d: non-empty piggyback payloads delivered           */
  SWIMSIM_CTR_RUMORS_SEEN = 6,    /* rumours examined by receivers                       */
  SWIMSIM_CTR_CHANGES = 7,        /* r: (observer,subject) view entries changed          */
  SWIMSIM_CTR_PB_WRITES = 8,      /* c: piggyback buffers rewritten                      */
  SWIMSIM_CTR_TIMERS_FIRED = 9,   /* Suspect -> Dead by timeout                          */
  SWIMSIM_CTR_REFUTES = 10,       /* incarnation bumps (src/Core.hs:155-166)             */
  SWIMSIM_CTR_EVENTS_DROPPED = 11,/* events lost to a full ring (counted before the per-(tick,
                                     observer, subject) collapse: once the ring overflows the
                                     exact count is implementation-defined, not protocol state) */
  SWIMSIM_CTR_ACTIVE_MEMBERS = 12,/* up-member ticks actually processed                  */
  SWIMSIM_CTR_EVDIGEST = 13,      /* running digest of every view / incarnation change   */
  SWIMSIM_CTR_FALSE_DEADS = 14,   /* ... of a member that was actually up: the false-positive
                                     Dead count of BASELINE config 5, per (observer, subject);
                                     "up" whatever its incarnation -- an observer burying the old
                                     incarnation of a member that has come back counts          */
  SWIMSIM_CTR_SETTLED = 15,       /* subjects settled (view columns reclaimed; gc_ticks) */
  SWIMSIM_CTR_EVICTED = 16,       /* member-map entries evicted (view_cap): in the map when the tick started, back at the
                                     default when it ended */
  SWIMSIM_CTR_COUNT = 17
};

#define SWIMSIM_TICK_NONE UINT64_MAX

/* ---- lifecycle ---------------------------------------------------------- */

/* Defaults = `parseConfig` (src/Util.hs:44-50): numToGossip 10, gossipInterval
 * 200 000 us; simulator fields zero (n_members must be set by the caller). */
int swimsim_default_config(swimsim_config_t* cfg);

/* Replaces `configure` + `makeStore` + `makeSelf` for all N members at once
 * (src/Util.hs:76-107): every member up, incarnation 0, every view "all Alive@0",
 * empty piggyback buffers, tick 0.  On failure *out is NULL and the message is
 * available via swimsim_last_error(NULL). */
int swimsim_create(const swimsim_config_t* cfg, swimsim_t** out);
/* The same with the failure text copied into the caller's buffer (NUL-terminated, truncated to errcap): for
 * hosts whose runtime may move a thread between two foreign calls (GHC's `safe` calls), where the per-thread
 * text of swimsim_last_error(NULL) could be somebody else's by the time it is read. */
int swimsim_create_msg(const swimsim_config_t* cfg, swimsim_t** out, char* err, size_t errcap);
void swimsim_destroy(swimsim_t* h);
const char* swimsim_last_error(const swimsim_t* h);

/* ---- fault injection (the simulator owns ground truth) */

/* Member `member` is down (up=0) / up again (up=1) FOR tick `tick` and after.
 * `tick` must be >= the current tick.  Going down loses the process's volatile
 * protocol state: its piggyback queue is dropped (the member map and its
 * suspicion deadlines are kept -- they stand for the state a restarted node
 * pulls from its join host).  Coming back up bumps the member's incarnation,
 * announces Alive (cause JOIN) and fires the suspicion deadlines that passed
 * while it was down. */
int swimsim_schedule_fault(swimsim_t* h, uint64_t tick, uint32_t member, uint8_t up);

/* ---- the hot path -------------------------------------------------------- */

/* Runs `nticks` protocol periods for every member: replaces the body of
 * `failureDetector`'s forever-loop (src/Core.hs:236-240), `probeNode'`
 * (:243-269), the receive side `handleUDPMessage.process` (:89-117), the state
 * rules `suspectOrDeadNode'`/`aliveNode` (:142-218) and `disseminate`'s
 * piggyback queue (:127-138), for all members at once.  Blocking. */
int swimsim_step(swimsim_t* h, uint32_t nticks);

int swimsim_tick(const swimsim_t* h, uint64_t* tick);

/* ---- results -------------------------------------------------------------- */

/* Membership events since the last drain, sorted by (tick, observer, subject);
 * several changes of one (tick, observer, subject) are collapsed to the final
 * one.  Replaces reading `Broadcast` values off `storeGossip` (src/Core.hs:280). */
int swimsim_drain_events(swimsim_t* h, swimsim_event_t* buf, size_t cap, size_t* n_out);

/* Non-default entries of `observer`'s member map, sorted by subject: replaces
 * `members` (src/Core.hs:76-77) / `readTVar storeMembers`.  Every member not
 * listed is Alive at incarnation 0 -- or was removed as Dead by settling. */
int swimsim_read_view(swimsim_t* h, uint32_t observer, swimsim_view_entry_t* buf,
                      size_t cap, size_t* n_out);

int swimsim_read_member(swimsim_t* h, uint32_t member, swimsim_member_t* out);

/* out[j] = first tick at which some live member's probe of j ended in Suspect
 * while j was down (SWIMSIM_TICK_NONE if never); n must be n_members. */
int swimsim_first_detect(swimsim_t* h, uint64_t* out, size_t n);

/* 64-bit digest of the complete semantic state (all views, incarnations,
 * piggyback buffers, live timers, first-detection ticks, tick counter).
 * Independent of internal slot assignment and processing order. */
int swimsim_digest(swimsim_t* h, uint64_t* out);

int swimsim_counters(swimsim_t* h, uint64_t* out, size_t n);

/* How far a rumour has got -- the "dissemination ticks-to-all" of BASELINE config 5 is the first tick at which
 * out[0] == out[1]:  out[1] = members that are up, `subject` itself not counted; out[0] = those of them whose entry
 * about `subject` is at least {incarnation, state} in the merge order of the state rule (src/Core.hs:142-187: higher
 * incarnation wins, then Dead > Suspect > Alive; an entry settling has folded into the base counts as that base).
 * Reads state between ticks, changes nothing.  On a sharded handle both numbers cover the members the handle owns
 * (the parts add up). */
int swimsim_coverage(swimsim_t* h, uint32_t subject, uint8_t state, uint32_t incarnation, uint64_t out[2]);

/* Occupancy of the bounded tables (what SWIMSIM_ERR_CAPACITY guards), for long runs:
 * out[0] = view rows ever handed out (high-water mark; bounded under settling), out[1] = subjects that hold a
 * row now (max_subjects bounds it), out[2] = reclaimed rows waiting for reuse, out[3] = rumour ids handed out
 * so far (the id counter wraps, SWIM_RID_BITS), out[4] = view rows allocated; with n >= 7 also out[5] = entries in the
 * fuller of the two inbox overflow lists, out[6] = their capacity.  n >= 5. */
int swimsim_table_stats(swimsim_t* h, uint64_t* out, size_t n);

/* ---- unit-level hooks (what test/Spec.hs exercises directly) --------------- */

/* `kRandomMembers store n excludes` (src/Core.hs:69-74; test/Spec.hs:108-139)
 * evaluated for `observer` on the device with the current tick's selection
 * stream.  Writes min(n, eligible) distinct member ids. */
int swimsim_k_random_members(swimsim_t* h, uint32_t observer, uint32_t n,
                             const uint32_t* excludes, size_t n_excludes,
                             uint32_t* out, size_t cap, size_t* n_out);

/* Overwrite one view entry (test fixture: the reference's tests swap a hand-built
 * map into `storeMembers`, test/Spec.hs:45-56,101). */
int swimsim_set_view(swimsim_t* h, uint32_t observer, uint32_t subject, uint8_t state,
                     uint32_t incarnation);

/* Resolved configuration (defaults filled in). */
int swimsim_get_config(const swimsim_t* h, swimsim_config_t* out);

/* ---- the outside world (the live-node bridge, include/swimbridge.h) ----
 * A Suspect / Alive / Dead message about `subject` that reaches simulated member `observer` from OUTSIDE the
 * simulation -- what `process` does with such a message from the socket (src/Core.hs:110-117): it is delivered
 * to `observer` in the next tick that is stepped, next to the rumours the tick's Pings and Acks carry (same state
 * rule, same events, re-gossiped if accepted) -- if `observer` is up when that tick starts and stays up through the
 * tick's scheduled changes; nobody listens otherwise.  The subject gets a view row like any subject somebody states a
 * rumour about.  On a sharded cluster the message goes to the handle that owns `observer` (SWIMSIM_ERR_INVALID on the others).  Not
 * available with bounded member maps.  A known limit: an observer that has MORE than inbox_cap messages pending, goes
 * down and comes back up within the one tick that delivers them keeps the part of them that sat in the inbox overflow list (the
 * oracle drops all of them); no test or workload gets there (inbox_cap messages to one observer between two ticks). */
int swimsim_inject_rumor(swimsim_t* h, uint32_t observer, uint32_t subject, uint8_t state, uint32_t incarnation);
/* Sharded clusters: the handles that do NOT own `observer` are told of the message too -- it opens the subject's view row, and a view
 * row is a property of the whole cluster (a state pull on any shard walks it in the tick the message is delivered). */
int swimsim_note_outside_rumor(swimsim_t* h, uint32_t observer, uint32_t subject);

/* ---- sharded clusters (one handle per GPU / process) -------------------------------
 * The population is split into n_shards contiguous id ranges.  Every shard gets the SAME
 * configuration (n_members = whole population) and the SAME fault schedule; ground truth and
 * probe outcomes need no communication, only piggyback payloads cross shards.  View rows and rumour ids
 * are per-shard numberings; what crosses shards is named by (subject, incarnation<<2|state).  Every shard
 * holds a REPLICA of what a delivery "dst merges src's queue" reads about src: its start-of-tick queue mask
 * (over its owner's ring of the tick) and a queue byte.  One tick (round 5) is
 *   phase1  -> round 1, an ALL-GATHER: the caller delivers to every peer p this shard's
 *                kind 0  send[0][0 .. counts[p]) -- ONE segment, the same for every peer --: 16-byte records, the tick's ring
 *                        dictionary (64 x {subject, incarnation<<2|state}, one per mask position = 32 records), then the
 *                        queues that travel as lists this tick (5 records each: {member, n, tick, -} + 8 x {subject, key}:
 *                        a queue with an entry outside the mask window, every queue in a tick after a burst of rumour ids)
 *                kinds 5 / 6  its slice of the queue masks (8 bytes per member) and queue bytes (1 byte)
 *                        (swimsim_shard_gather_buffers), into peer p's recv[..][me]
 *   phase2  -> round 2, an all-to-all-v: send[1][p][0 .. counts[n_shards + p]) -- 8-byte records {dst, src} "member dst (of
 *                shard p) merges the start-of-tick queue of member src" -- to shard p's recv[1][me][..]
 *   phase3
 * on every shard in lock step (swim_amd/shard.py does the exchange with torch.distributed: RCCL over xGMI on
 * GPUs).  Nothing is requested and nothing is answered: the probe outcomes, the Acks' payloads (read from the replica
 * through the owner's dictionary) and every filter are local.  Buffers are owned by the library; kind 2 (72-byte explicit
 * payloads before round 5) no longer exists: shard_info reports x_cap = 0, shard_buffers NULL.
 * counts[] arrays hold n_shards entries per kind (kind-major, 3 * n_shards).  swimsim_step is refused
 * on sharded handles; digest / counters / events return this shard's part (the parts add up /
 * concatenate); view and member reads are answered by the owner only; first-detection ticks must be
 * combined (element-wise minimum) and set back before digest or first_detect are read. */
/* Bounded handles (view_cap > 0) shard the same way and are stepped by the same calls, with a simpler exchange:
 *   phase1  the tick's scheduled changes; this shard's slice of two replicated tables is ready to be ALL-GATHERED in round 1
 *           (swimsim_shard_gather_buffers): everybody's start-of-tick queue line (64-byte records, kind 5) and member byte
 *           (1 byte: up, queue length; kind 6).  No records of kind 0 (shard_info reports r_cap = x_cap = 0).
 *   phase2  one period of failureDetector for this shard's members; round 2 carries 16-byte records {dst, src, -, -} of kind 1:
 *           "dst merges src's queue" for members dst of the peer (the receiver appends src to dst's inbox and reads src's line
 *           from its replica).  Nothing of kind 2.
 *   phase3  end of tick.
 * swimsim_shard_step drives it like a dense cluster: xchg(ctx, 1, ..) gathers (kind-5 counts at
 * [n_shards + p], kind-6 at [2 n_shards + p], n_local each), xchg(ctx, 2, ..) delivers the kind-1 records. */
#define SWIMSIM_RREC_BYTES 16u
#define SWIMSIM_PREC_BYTES 8u            /* dense handles; bounded handles: */
#define SWIMSIM_PREC_BOUNDED_BYTES 16u
int swimsim_shard_info(const swimsim_t* h, uint32_t* lo, uint32_t* n_local, uint32_t* r_cap,
                       uint32_t* p_cap, uint32_t* x_cap);
int swimsim_shard_buffers(swimsim_t* h, void** send /*[3]*/, void** recv /*[3]*/);
int swimsim_shard_phase1(swimsim_t* h, uint32_t* counts /*[3*n_shards] out*/);
int swimsim_shard_phase2(swimsim_t* h, const uint32_t* r_counts_in /*[n_shards]*/,
                         uint32_t* counts /*[3*n_shards] out: kinds 1 and 2 now final*/);
int swimsim_shard_phase3(swimsim_t* h, const uint32_t* p_counts_in, const uint32_t* x_counts_in);
/* The same tick driven by the library: `nticks` times phase1 -> xchg(round 1) -> phase2 -> xchg(round 2) ->
 * phase3.  `xchg` is the embedder's exchange: for every peer p != me and every kind k of the round
 * (round 1: kind 0 at [p] + the gathered kinds 5 / 6 at [n_shards + p], [2 n_shards + p]; round 2: kind 1) it delivers
 * counts_out[..] records to peer p's recv[k][me][..] and writes into counts_in[..] how many records arrived from p (buffers:
 * swimsim_shard_buffers / swimsim_shard_gather_buffers; record sizes: SWIMSIM_*REC_BYTES; kind 0 and the gathered kinds send
 * ONE segment to every peer).  It returns 0, or a non-zero value that aborts the
 * step with SWIMSIM_ERR_STATE.  Every shard of the cluster must make the same call; this is what a host
 * without swim_amd/shard.py binds (haskell/Swim/Sim.hs: stepShard) -- MPI_Alltoallv, RCCL send/recv or, as
 * swim_amd/shard.py does, torch.distributed. */
/* A cluster whose handles live in ONE process -- one per GPU, or several on one GPU: `nticks` periods with the tick loop AND the
 * exchange inside the library, no host synchronisation between the first tick and the last.  hs[k] = shard k of n handles of ONE
 * configuration (checked field by field) with one fault schedule.  Dense handles (round 5): the kernels read the peers' send
 * buffers WHERE THEY LIE (same device, or a peer device over xGMI: hipDeviceEnablePeerAccess, refused with SWIMSIM_ERR_DEVICE
 * where the devices cannot) with the counts from the peers' own words, ordered by events on the handles' streams; only the
 * replica slices are copied.  Every option of a sharded handle included: settling (round 3), state pulls (round 0: join_pull, pull_ticks),
 * messages from outside (swimsim_inject_rumor).  Bounded handles
 * (view_cap): the all-gather and the all-to-all-v of DESIGN.md 6 as device-to-device (peer) copies.
 * (Multi-process clusters use swimsim_shard_step with the embedder's exchange.) */
int swimsim_cluster_step(swimsim_t** hs, uint32_t n, uint32_t nticks);
typedef int (*swimsim_exchange_fn)(void* ctx, int round, const uint32_t* counts_out /*[3*n_shards]*/,
                                   uint32_t* counts_in /*[3*n_shards]*/);
int swimsim_shard_step(swimsim_t* h, uint32_t nticks, swimsim_exchange_fn xchg, void* ctx);
/* Settling on a sharded cluster (gc_ticks > 0): a subject settles when it is quiet on EVERY shard, and every shard
 * must commit the same base in the same tick.  After phase3 each shard holds a list of 8-byte records
 * {subject | flags, largest entry among its up members} about its rows (kind 3: the SAME list for every peer, at
 * send[p][0 .. counts[p]) of swimsim_shard_settle_buffers); round 3 delivers it to every peer's recv[me][..] like
 * the other rounds, and swimsim_shard_settle_commit ends the tick.  swimsim_shard_step does it by itself and
 * calls xchg(ctx, 3, counts_out, counts_in) with the kind-3 counts at index [p]. */
#define SWIMSIM_SREC_BYTES 8u
#define SWIMSIM_JREC_BYTES 16u
int swimsim_shard_settle_buffers(swimsim_t* h, void** send, void** recv, uint32_t* cap /* records per peer segment */);
int swimsim_shard_settle_counts(swimsim_t* h, uint32_t* counts /*[n_shards] out*/);
int swimsim_shard_settle_commit(swimsim_t* h, const uint32_t* counts_in /*[n_shards]*/);
/* join_pull on a sharded cluster: a join host may live on another shard than the member that comes up, and the pull
 * precedes the tick's probes.  swimsim_shard_phase0 applies the tick's faults; when members come up in this tick
 * (the schedule is replicated: every shard gets the same answer) it sets *round_needed and the owners of the hosts
 * have written 16-byte records {joiner, subject, the host's entry, -} (kind 4) into send[p][0 .. counts[p]) of
 * swimsim_shard_join_buffers: round 0 delivers them, swimsim_shard_join_ingest reports the arrivals, phase1
 * continues the tick.  A no-op when join_pull is off or the tick has no joins; swimsim_shard_step does all of it
 * and calls xchg(ctx, 0, ..) with the kind-4 counts at index [p]. */
int swimsim_shard_phase0(swimsim_t* h, uint32_t* counts /*[n_shards] out*/, int* round_needed);
int swimsim_shard_join_buffers(swimsim_t* h, void** send, void** recv, uint32_t* cap /* records per peer segment */);
int swimsim_shard_join_ingest(swimsim_t* h, const uint32_t* counts_in /*[n_shards]*/);
/* The replicas of a dense shard (and of a bounded one: queue lines instead of masks): after phase1 every shard's slice of two
 * tables -- 8 bytes (kind 5) and 1 byte (kind 6) per member, n_local records each -- is all-gathered: send[k] is this shard's
 * slice (the SAME bytes go to every peer), recv[k] the whole table, peer p's slice at recv[k] + p * n_local * record size.  It
 * travels WITH round 1: swimsim_shard_step's xchg(ctx, 1, ..) finds the kind-5 counts at [n_shards + p] and the kind-6 counts at
 * [2 n_shards + p] (n_local for every peer) next to the kind-0 counts at [p]. */
#define SWIMSIM_GREC5_BYTES 8u          /* dense handles; bounded handles gather 64-byte queue lines: */
#define SWIMSIM_GREC5_BOUNDED_BYTES 64u
#define SWIMSIM_GREC6_BYTES 1u
int swimsim_shard_gather_buffers(swimsim_t* h, void** send /*[2]*/, void** recv /*[2]*/, uint32_t* n_local);
/* What this shard put on the wire in the LAST tick it stepped (for the bench's xGMI figure): out[0] = bytes of round 1 it publishes
 * to EACH peer (replica slices + dictionary + lists), out[1] = bytes of round-2 records it sent to all peers together, out[2] =
 * round-2 records it kept (deliveries its own ingest completes), out[3] = queues that travelled as lists. */
int swimsim_shard_traffic(swimsim_t* h, uint64_t out[4]);
int swimsim_shard_get_first_suspect(swimsim_t* h, uint32_t* out, size_t n);
int swimsim_shard_set_first_suspect(swimsim_t* h, const uint32_t* combined, size_t n);

/* ---- measurement ------------------------------------------------------------ */

/* When enabled, swimsim_step brackets every tick-kernel launch with HIP events on the
 * library's own stream (torch.cuda.Event would not see it) and accumulates the elapsed
 * times.  out[0] = total ms in probe_kernel, out[1] = total ms in merge_kernel,
 * out[2] = number of ticks measured; reading resets nothing, enabling resets. */
int swimsim_kernel_timing_enable(swimsim_t* h, int enable);
int swimsim_kernel_timing(swimsim_t* h, double* out, size_t n);

#ifdef __cplusplus
}
#endif
#endif /* SWIMSIM_H */
