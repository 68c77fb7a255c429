/*
 * swim_oracle.c -- CPU ORACLE (test infrastructure, not product; see swim_oracle.h).
 *
 * Message-level restatement of jpfuentes2/swim's gossip / failure-detection
 * round as a bulk-synchronous tick.  Every function cites the reference lines it
 * follows; "Dn" refers to the defect/divergence register in DESIGN.md section 3
 * (= SURVEY.md appendix A).  Paths are relative to the reference checkout.
 *
 * Tick semantics, for tick t:
 *   0. faults scheduled for t are applied (ground truth up[]).
 *   1. every up member runs one period of `failureDetector` (src/Core.hs:233-241):
 *      picks P targets (`kRandomMembers`), probes each (`probeNode'`), and on a
 *      missing ack asks K proxies (`IndirectPing`).  Messages are delivered
 *      in-memory through `process` (src/Core.hs:89-117) subject to the loss hash
 *      and to up[dst].  Every message carries the sender's START-OF-TICK
 *      piggyback buffer as a compound envelope (src/Types.hs:96-119).
 *   2. all state changes land at the end of the tick (Jacobi): per member, in
 *      this phase order: suspicion timers, own failed probes, received rumours;
 *      each applied with the commutative merge `max (incarnation, state)`.
 *   3. changed entries become rumours in the member's piggyback buffer.
 *   4. settling (gc_ticks; `removeDeadNodes`, src/Core.hs:65-67): subjects nobody has changed its mind
 *      about for G ticks are reconciled and their view columns reclaimed (settle()).
 * The outcome of a tick does not depend on the order in which members or
 * messages are processed (property-tested via swimoracle_set_shuffle).
 */
#include "swim_oracle.h"

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* spec constants                                                             */
/* ------------------------------------------------------------------------- */
#define PB_SLOTS 8          /* piggyback buffer slots per member (D5)              */
#define SEL_ATTEMPTS 8      /* rejection-sampling attempts per pick (H4)           */
#define INC_MAX 0x3FFFFFu   /* incarnation must fit 22 bits (key = inc<<2|state)   */

enum { P_SELECT = 1, P_PROXY = 2, P_L_PING = 3, P_L_ACK = 4, P_L_REQ = 5,
       P_L_FWD = 6, P_L_BACK = 7, P_L_RELAY = 8, P_JOIN = 9, P_PULL = 10 };

enum { TAG_SELF = 0x53454c46u, TAG_VIEW = 0x56494557u, TAG_PB = 0x50425546u,
       TAG_TIMER = 0x54494d52u, TAG_FD = 0x46444554u, TAG_EV = 0x45564e54u,
       TAG_INC = 0x494e4352u, TAG_TICK = 0x5449434bu, TAG_MEMBER = 0x4d454d42u,
       TAG_BASE = 0x42415345u };

#define NONE32 0xFFFFFFFFu

static inline uint32_t key_make(uint32_t inc, uint32_t st) { return (inc << 2) | st; }
static inline uint32_t key_inc(uint32_t k) { return k >> 2; }
static inline uint32_t key_state(uint32_t k) { return k & 3u; }

/* ------------------------------------------------------------------------- */
/* hashes.  Replaces the global StdGen of src/Util.hs:40 (F7).  */
/* ------------------------------------------------------------------------- */
static inline uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
static inline uint32_t tick_key(uint64_t seed, uint32_t t) {
  return mix32((uint32_t)seed + mix32((uint32_t)(seed >> 32) + mix32(t + 0x9E3779B9u)));
}
static inline uint32_t hash_h(uint32_t tk, uint32_t a, uint32_t b, uint32_t c) {
  return mix32(mix32(mix32(tk ^ a) + b) ^ c);
}
uint32_t swimoracle_hash(uint64_t seed, uint32_t tick, uint32_t a, uint32_t b, uint32_t c) {
  return hash_h(tick_key(seed, tick), a, b, c);
}
static inline uint64_t mix64(uint64_t x) {
  x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull; x ^= x >> 27; x *= 0x94d049bb133111ebull;
  x ^= x >> 31;
  return x;
}
static inline uint64_t h4(uint64_t tag, uint64_t a, uint64_t b, uint64_t c) {
  return mix64(mix64(mix64(mix64(tag) + a) + b) + c);
}
/* A view change of member i about subject s in tick t moves the running event digest by
 * E(t, i, s) * (new key - old key), E = h3(TAG_EV, t << 32 | i, s) | 1: linear in the key, so several changes of one
 * entry in one tick telescope to first -> last whatever the order they are applied in. */
static inline uint64_t ev_weight(uint32_t t, uint32_t i, uint32_t s) {
  return mix64(mix64(mix64((uint64_t)TAG_EV) + (((uint64_t)t << 32) | i)) + s) | 1ull;
}

/* ------------------------------------------------------------------------- */
/* state                                                                      */
/* ------------------------------------------------------------------------- */
typedef struct { uint32_t subject, key; uint8_t tx; } orumor_t;
typedef struct { int n; orumor_t r[PB_SLOTS]; } opb_t;           /* disseminate's queue */
typedef struct { uint32_t subject, deadline; } otimer_t;
typedef struct { otimer_t* v; uint32_t head, n, cap; } otimerq_t; /* FIFO               */
typedef struct { uint32_t key, since1; } oentry_t;  /* Member: state+inc, lastChange+1 */
typedef struct { uint32_t dst, subject, key; } opend_t;
typedef struct { uint32_t i, j; } ofail_t;
typedef struct { uint32_t tick, member; uint8_t up; uint32_t order; } ofault_t;

/* internal message = public message + routing context the tick model needs     */
typedef struct {
  swimoracle_msg_t m;
  uint32_t relay_to;  /* proxy must relay the ack to this requester (D9)           */
  uint32_t lidx;      /* probe index (p, or p<<8|k) keying the loss hash           */
  int via_proxy;      /* hop class: 0 = direct ping/ack, 1 = proxied               */
} omsg_t;

typedef struct { opend_t* v; size_t n, cap; } opendv_t;
/* bounded member maps (view_cap; include/swimsim.h "Bounded member maps"): a member's exceptions to the default, sorted by
 * subject -- `Map String Member` (src/Types.hs:55) with a capacity */
typedef struct { uint32_t subject, key, since1; } osent_t;
typedef struct { osent_t* v; uint32_t n; } otab_t;

/* Worker context.  A tick is two parallel phases over contiguous member ranges (Jacobi semantics make
 * both embarrassingly parallel): A = every up member's failureDetector period (reads start-of-tick
 * state only; delivered rumours are appended to per-destination-range lists), B = every member's end of
 * tick (touches only its own state).  One worker = the single-threaded oracle; swimoracle_set_threads
 * adds workers for the CPU baseline and for oracle-checked runs at full size.  Results do not depend on
 * the number of workers (tests/test_oracle_semantics.py). */
typedef struct octx {
  struct swimoracle* o;
  uint32_t w, lo, hi;       /* worker index, member range [lo, hi) */
  uint64_t counters[SWIMSIM_CTR_COUNT];
  opendv_t* out;            /* [nworkers] rumours delivered this tick, by destination range */
  ofail_t* fails; size_t nfails, fails_cap;
  swimsim_event_t* events; size_t nevents, events_cap;
  opend_t* sorted; size_t sorted_cap; uint32_t* off;   /* phase B: this range's rumours by receiver */
  int ctx_acked;            /* probe context (processing is synchronous, depth first) */
  char pad_[128];           /* workers' contexts sit in one array: keep their hot counters on separate cache lines */
} __attribute__((aligned(128))) octx_t;

struct swimoracle {
  swimsim_config_t cfg;     /* resolved */
  uint32_t N, P, K, S, L, loss_thr;
  uint64_t tick;
  uint32_t tk;              /* tick key of the running tick */
  uint8_t* up;
  uint32_t* self_inc;       /* storeIncarnation (src/Types.hs:54) */
  opb_t* pb;
  otimerq_t* timers;
  uint8_t* nsent;
  /* sparse views: slot-major columns over the subjects anyone ever gossiped about */
  uint32_t* slot_of;        /* subject -> slot+1, 0 = none (view entry = the default)   */
  uint32_t* subject_of;
  oentry_t** cols;
  uint32_t nslots, slots_cap;
  uint32_t nlive;           /* columns in use (max_subjects bounds this)               */
  uint32_t* free_at;        /* [slot] NONE32 = in use, else the tick from which the column may be reused */
  /* settling (gc_ticks): the default entry about s is base[s] (0 = Alive@0), settled at base_since[s] */
  uint32_t G;
  uint32_t* base; uint32_t* base_since;
  uint32_t* last_change;    /* [subject] last tick any view entry about it changed / it announced itself */
  uint32_t C;               /* view_cap: 0 = unbounded (columns above), else every member keeps <= C entries in tab[] */
  otab_t* tab;              /* [N] (view_cap > 0) */
  int literal_rule;         /* oracle-only: apply the LITERAL suspectOrDeadNode' instead of the merge */
  uint64_t d13_hits;        /* proposals on which the literal rule and the merge disagree            */
  /* fault schedule */
  ofault_t* faults; size_t nfaults, faults_cap; uint32_t fault_order;
  uint32_t* first_suspect;  /* NONE32 = never */
  uint32_t* crash_tick;
  /* workers */
  octx_t* ctx; uint32_t nworkers, chunk;      /* member m belongs to worker m / chunk */
  pthread_t* threads; pthread_barrier_t bar; int phase; int quit;
  pthread_mutex_t mu;       /* slot allocation, error text */
  /* events */
  swimsim_event_t* events; size_t nevents, events_cap_alloc;
  uint64_t counters[SWIMSIM_CTR_COUNT];
  uint64_t shuffle_seed;
  /* capture mode (unit-level hook swimoracle_process) */
  int capture; int capture_literal_d8;
  swimoracle_msg_t* cap_out; size_t cap_cap, cap_n;
  /* rumours from outside the simulation (swimsim_inject_rumor): delivered in the next tick */
  opend_t* inj; size_t ninj, inj_cap;
  int poisoned;
  char err[256];
};

static char g_create_err[256];

static int fail(swimoracle_t* o, int code, const char* msg) {
  if (o) {
    pthread_mutex_lock(&o->mu);
    snprintf(o->err, sizeof o->err, "%s", msg);
    if (code == SWIMSIM_ERR_CAPACITY) __atomic_store_n(&o->poisoned, 1, __ATOMIC_RELAXED);
    pthread_mutex_unlock(&o->mu);
  } else snprintf(g_create_err, sizeof g_create_err, "%s", msg);
  return code;
}

const char* swimoracle_last_error(const swimoracle_t* h) { return h ? h->err : g_create_err; }

static uint32_t ceil_log2(uint64_t x) { uint32_t r = 0; while ((1ull << r) < x) r++; return r; }

/* `parseConfig` (src/Util.hs:44-50) */
int swimoracle_default_config(swimsim_config_t* cfg) {
  if (!cfg) return SWIMSIM_ERR_INVALID;
  memset(cfg, 0, sizeof *cfg);
  cfg->struct_size = (uint32_t)sizeof *cfg;
  cfg->abi_version = SWIMSIM_ABI_VERSION;
  cfg->num_to_gossip = 10;            /* src/Util.hs:48 */
  cfg->gossip_interval_us = 200000;   /* src/Util.hs:49 `milliseconds 200` */
  return SWIMSIM_OK;
}

/* ------------------------------------------------------------------------- */
/* views                                                                      */
/* ------------------------------------------------------------------------- */
/* member i's entry about s; the default (no column, or an untouched cell) is the settled base --
 * Alive@0 until s is settled for the first time */
static inline const osent_t* tab_find(const otab_t* tb, uint32_t s) {
  uint32_t lo = 0, hi = tb->n;                       /* sorted by subject */
  while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (tb->v[mid].subject < s) lo = mid + 1; else hi = mid; }
  return (lo < tb->n && tb->v[lo].subject == s) ? &tb->v[lo] : NULL;
}
static inline oentry_t view_get(const swimoracle_t* o, uint32_t i, uint32_t s) {
  oentry_t z = {o->base[s], 0};
  if (o->C) {                                        /* bounded map: an entry, or the default */
    const osent_t* e = tab_find(&o->tab[i], s);
    if (e) { oentry_t r = {e->key, e->since1}; return r; }
    return z;
  }
  uint32_t sl = __atomic_load_n(&o->slot_of[s], __ATOMIC_ACQUIRE);
  if (!sl) return z;
  oentry_t e = o->cols[sl - 1][i];
  return e.key ? e : z;
}

/* cell of member i about subject s, the subject's column created on first use (cols / subject_of are
 * allocated once for every possible column, so readers never see them move).  A cell with key 0 is
 * "default": callers compare against view_get. */
static oentry_t* view_ref(swimoracle_t* o, uint32_t i, uint32_t s) {
  uint32_t sl = __atomic_load_n(&o->slot_of[s], __ATOMIC_ACQUIRE);
  if (!sl) {
    pthread_mutex_lock(&o->mu);
    sl = o->slot_of[s];
    if (!sl && o->nlive < o->cfg.max_subjects) {
      uint32_t t = (uint32_t)o->tick, pick = NONE32;
      for (uint32_t r = 0; r < o->nslots; r++)             /* a reclaimed column, if one is reusable */
        if (o->free_at[r] != NONE32 && o->free_at[r] <= t) { pick = r; break; }
      if (pick == NONE32 && o->nslots < o->slots_cap) {
        oentry_t* col = (oentry_t*)calloc(o->N, sizeof(oentry_t));
        if (col) { pick = o->nslots++; o->cols[pick] = col; }
      }
      if (pick != NONE32) {
        o->free_at[pick] = NONE32;
        o->subject_of[pick] = s;
        o->nlive++;
        sl = pick + 1;
        __atomic_store_n(&o->slot_of[s], sl, __ATOMIC_RELEASE);
      }
    }
    pthread_mutex_unlock(&o->mu);
    if (!sl) { fail(o, SWIMSIM_ERR_CAPACITY, "max_subjects exceeded"); return NULL; }
  }
  return &o->cols[sl - 1][i];
}

/* `isAlive` (src/Core.hs:33-34) on member i's view of s */
static inline int is_alive_in_view(const swimoracle_t* o, uint32_t i, uint32_t s) {
  return key_state(view_get(o, i, s).key) == SWIMSIM_ALIVE;
}

/* ------------------------------------------------------------------------- */
/* kRandomMembers (src/Core.hs:69-74) + shuffle (src/Util.hs:37-42)            */
/*                                                                             */
/* Reference: n first elements of a uniform shuffle of the members that are    */
/* alive IN THE LOCAL VIEW and not in `excludes`.  Equivalent in distribution: */
/* n draws without replacement.  Here: rejection sampling with the counter     */
/* RNG (H4): per pick SEL_ATTEMPTS draws, then a deterministic cyclic scan so  */
/* that "fewer than n candidates => all of them" holds exactly                 */
/* (test/Spec.hs:117-128).  Self is never eligible (D15).                      */
/* ------------------------------------------------------------------------- */
static int eligible(const swimoracle_t* o, uint32_t i, uint32_t c, const uint32_t* excl,
                    size_t nexcl, const uint32_t* picks, uint32_t npicks) {
  if (c == i) return 0;                                   /* D15 */
  for (size_t e = 0; e < nexcl; e++) if (excl[e] == c) return 0;  /* notElem m excludes */
  for (uint32_t e = 0; e < npicks; e++) if (picks[e] == c) return 0;
  return is_alive_in_view(o, i, c);                       /* isAlive m */
}

static uint32_t k_random_members(const swimoracle_t* o, uint32_t i, uint32_t n,
                                 const uint32_t* excl, size_t nexcl, uint32_t purpose,
                                 uint32_t hi_idx, uint32_t* out) {
  uint32_t np = 0;
  for (uint32_t p = 0; p < n; p++) {
    uint32_t c = 0; int found = 0;
    uint32_t base = (purpose == P_SELECT) ? (p << 8) : ((hi_idx << 16) | (p << 8));
    for (uint32_t a = 0; a < SEL_ATTEMPTS; a++) {
      uint32_t r = hash_h(o->tk, i, (purpose << 24) | base | a, 0);
      c = (uint32_t)(((uint64_t)r * o->N) >> 32);
      if (eligible(o, i, c, excl, nexcl, out, np)) { found = 1; break; }
    }
    if (!found) {
      uint32_t c0 = c + 1 == o->N ? 0 : c + 1;
      for (uint32_t d = 0; d < o->N; d++) {
        c = c0 + d; if (c >= o->N) c -= o->N;
        if (eligible(o, i, c, excl, nexcl, out, np)) { found = 1; break; }
      }
    }
    if (!found) break;      /* fewer than n candidates: take what exists */
    out[np++] = c;
  }
  return np;
}

/* ------------------------------------------------------------------------- */
/* network: loss + delivery + process                                         */
/* ------------------------------------------------------------------------- */
static inline int lost(const swimoracle_t* o, uint32_t purpose, uint32_t src, uint32_t dst, uint32_t idx) {
  if (!o->loss_thr) return 0;
  return hash_h(o->tk, src, (purpose << 24) | idx, dst) < o->loss_thr;
}

static void pend_add(octx_t* c, uint32_t dst, uint32_t subject, uint32_t key) {
  opendv_t* q = &c->out[dst / c->o->chunk];
  if (q->n == q->cap) {
    q->cap = q->cap ? q->cap * 2 : 1024;
    q->v = (opend_t*)realloc(q->v, q->cap * sizeof *q->v);
  }
  q->v[q->n].dst = dst; q->v[q->n].subject = subject; q->v[q->n].key = key;
  q->n++;
}

static void process(octx_t* c, uint32_t self, uint32_t sender, const omsg_t* msg);

/* A datagram src -> dst: the control message plus the sender's start-of-tick
 * piggyback buffer as a compound envelope (src/Types.hs:96-119; D5, D11). */
static int deliver(octx_t* c, uint32_t purpose, uint32_t src, uint32_t dst, const omsg_t* msg) {
  swimoracle_t* o = c->o;
  if (lost(o, purpose, src, dst, msg->lidx)) return 0;
  if (!o->up[dst]) return 0;                      /* nobody listening */
  const opb_t* pb = &o->pb[src];
  if (pb->n > 0) {
    c->counters[SWIMSIM_CTR_PAYLOADS]++;
    for (int s = 0; s < pb->n; s++) {             /* handleUDPMessage: CC.concat over the envelope */
      omsg_t r; memset(&r, 0, sizeof r);
      uint32_t st = key_state(pb->r[s].key);
      r.m.type = st == SWIMSIM_SUSPECT ? SWIMO_MSG_SUSPECT : st == SWIMSIM_DEAD ? SWIMO_MSG_DEAD : SWIMO_MSG_ALIVE;
      r.m.incarnation = key_inc(pb->r[s].key);
      r.m.node = pb->r[s].subject;
      r.relay_to = NONE32;
      c->counters[SWIMSIM_CTR_RUMORS_SEEN]++;
      process(c, dst, src, &r);
    }
  }
  process(c, dst, src, msg);
  return 1;
}

static void emit(swimoracle_t* o, const swimoracle_msg_t* m) {
  if (o->cap_n < o->cap_cap) o->cap_out[o->cap_n] = *m;
  o->cap_n++;
}

static int accept_key(octx_t* c, uint32_t i, uint32_t s, uint32_t key, uint8_t cause,
                      uint32_t* refute_inc, opb_t* cand, uint32_t self_inc_start);

/* `process sender msg` (src/Core.hs:89-117) */
static void process(octx_t* c, uint32_t self, uint32_t sender, const omsg_t* msg) {
  swimoracle_t* o = c->o;
  switch (msg->m.type) {
    case SWIMO_MSG_ACK:                                    /* src/Core.hs:92-94 */
      if (o->capture) return;                              /* `return []` */
      if (msg->relay_to != NONE32) {                       /* proxy relays the ack (D9) */
        omsg_t a = *msg; a.relay_to = NONE32;
        deliver(c, P_L_RELAY, self, msg->relay_to, &a);
      } else {
        c->ctx_acked = 1;                                  /* invokeAckHandler (src/Core.hs:220-221) */
      }
      return;
    case SWIMO_MSG_PING:                                   /* src/Core.hs:97-101 */
      if (msg->m.node == self) {
        omsg_t a; memset(&a, 0, sizeof a);
        a.m.type = SWIMO_MSG_ACK; a.m.seq_no = msg->m.seq_no; a.m.to = sender;
        a.relay_to = msg->relay_to; a.lidx = msg->lidx; a.via_proxy = msg->via_proxy;
        if (o->capture) { emit(o, &a.m); return; }
        deliver(c, msg->via_proxy ? P_L_BACK : P_L_ACK, self, sender, &a);
      }
      return;                                              /* not for us: [] */
    case SWIMO_MSG_INDIRECT_PING: {                        /* src/Core.hs:105-108 */
      omsg_t p; memset(&p, 0, sizeof p);
      p.m.type = SWIMO_MSG_PING; p.m.node = msg->m.node; p.m.to = msg->m.target;
      if (o->capture && o->capture_literal_d8) {
        /* literal: `nextIncarnation store >>= \next -> Ping (fromIntegral next)` (D8) */
        o->self_inc[self] += 1;
        p.m.seq_no = o->self_inc[self];
      } else {
        p.m.seq_no = msg->m.seq_no;                        /* D8: relay the requester's seqNo */
      }
      p.relay_to = sender; p.lidx = msg->lidx; p.via_proxy = 1;
      if (o->capture) { emit(o, &p.m); return; }
      deliver(c, P_L_FWD, self, msg->m.target, &p);
      return;
    }
    case SWIMO_MSG_SUSPECT:                                /* src/Core.hs:110-117 */
    case SWIMO_MSG_DEAD:
    case SWIMO_MSG_ALIVE: {
      uint32_t st = msg->m.type == SWIMO_MSG_SUSPECT ? SWIMSIM_SUSPECT
                  : msg->m.type == SWIMO_MSG_DEAD ? SWIMSIM_DEAD : SWIMSIM_ALIVE;
      uint32_t key = key_make(msg->m.incarnation, st);
      if (o->capture) {                                    /* maybeBroadcast, applied now */
        uint32_t refute = NONE32; opb_t cand = o->pb[self];
        int ch = accept_key(c, self, msg->m.node, key, SWIMSIM_CAUSE_GOSSIP, &refute, &cand, o->self_inc[self]);
        if (refute != NONE32) {                            /* src/Core.hs:155-166 */
          o->self_inc[self] = refute + 1;
          swimoracle_msg_t a; memset(&a, 0, sizeof a);
          a.type = SWIMO_MSG_ALIVE; a.incarnation = o->self_inc[self]; a.node = self; a.broadcast = 1;
          emit(o, &a);
        } else if (ch) {
          swimoracle_msg_t b = msg->m; b.broadcast = 1; emit(o, &b);
        }
        o->pb[self] = cand;
        return;
      }
      pend_add(c, self, msg->m.node, key);                 /* lands at end of tick */
      return;
    }
    default: return;
  }
}

/* ------------------------------------------------------------------------- */
/* active side: failureDetector / probeNode' (src/Core.hs:233-269)             */
/* ------------------------------------------------------------------------- */
static void fails_add(octx_t* c, uint32_t i, uint32_t j) {
  if (c->nfails == c->fails_cap) {
    c->fails_cap = c->fails_cap ? c->fails_cap * 2 : 256;
    c->fails = (ofail_t*)realloc(c->fails, c->fails_cap * sizeof *c->fails);
  }
  c->fails[c->nfails].i = i; c->fails[c->nfails].j = j; c->nfails++;
}

/* probeNode' store currSeqNo m (src/Core.hs:243-254) */
static void probe_node(octx_t* c, uint32_t i, uint32_t p, uint32_t j) {
  swimoracle_t* o = c->o;
  omsg_t ping; memset(&ping, 0, sizeof ping);
  ping.m.type = SWIMO_MSG_PING; ping.m.seq_no = (uint32_t)o->tick + 1; ping.m.node = j; ping.m.to = j;
  ping.relay_to = NONE32; ping.lidx = p; ping.via_proxy = 0;
  c->counters[SWIMSIM_CTR_PINGS]++;
  c->ctx_acked = 0;
  deliver(c, P_L_PING, i, j, &ping);                       /* yield Direct (Ping ...) :246 */
  if (c->ctx_acked) return;                                /* unlessAck (D2, D3)          */
  c->counters[SWIMSIM_CTR_DIRECT_FAILED]++;
  /* kRandomMembers store (numToGossip cfg) [] :249 -- D7: exclude the target */
  uint32_t qs[256];
  uint32_t nq = k_random_members(o, i, o->K, &j, 1, P_PROXY, p, qs);
  int acked = 0;
  for (uint32_t k = 0; k < nq; k++) {                      /* yieldMany ... indirectPing :250 */
    omsg_t ip; memset(&ip, 0, sizeof ip);
    ip.m.type = SWIMO_MSG_INDIRECT_PING; ip.m.seq_no = (uint32_t)o->tick + 1;
    ip.m.target = j; ip.m.node = j;                        /* D12: node = member id        */
    ip.relay_to = NONE32; ip.lidx = (p << 8) | k; ip.via_proxy = 1;
    c->counters[SWIMSIM_CTR_PING_REQS]++;
    c->ctx_acked = 0;
    deliver(c, P_L_REQ, i, qs[k], &ip);
    acked |= c->ctx_acked;                                 /* any of the k relays (D9)     */
  }
  if (acked) return;                                       /* second unlessAck :251        */
  /* suspectNode store (Suspect (memberIncarnation m) name) :253 -- lands at end of tick */
  fails_add(c, i, j);
  c->counters[SWIMSIM_CTR_SUSPECTS]++;
  if (o->up[j]) c->counters[SWIMSIM_CTR_FALSE_SUSPECTS]++;
  else {                                                   /* min over the probers of j (any order) */
    uint32_t t = (uint32_t)o->tick, cur = __atomic_load_n(&o->first_suspect[j], __ATOMIC_RELAXED);
    while ((cur == NONE32 || t < cur) &&
           !__atomic_compare_exchange_n(&o->first_suspect[j], &cur, t, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  }
}

/* The "robust scheme" the reference asks for (FIXME at src/Core.hs:232): the round-robin target
 * selection of the SWIM paper (section 4.3) as a population-wide rotation (include/swimsim.h,
 * DESIGN.md section 8).  Rounds of R = ceil((N-1)/P) periods; round r uses a pseudo-random permutation
 * pi_r of 0..N-2 and probe p of period u of the round has offset 1 + pi_r(u*P + p) (no probe once
 * u*P + p >= N-1).  pi_r = a keyed bijection on ceil(log2(N-1))-bit words (xor, odd multiplications,
 * xor-shifts, one addition: each step is invertible) restricted to [0, N-1) by cycle walking.  An
 * arithmetic progression of offsets would be a permutation too, but its sums barely expand and the
 * gossip then spreads polynomially instead of exponentially. */
static uint32_t perm_bits(uint32_t x, uint32_t k1, uint32_t k2, uint32_t bits) {
  const uint32_t mask = bits >= 32 ? 0xFFFFFFFFu : (1u << bits) - 1u, sh = bits / 2 ? bits / 2 : 1;
  x = (x ^ k1) & mask;
  x = (x * 0x9E3779B1u) & mask; x ^= x >> sh;
  x = (x * 0x85EBCA6Bu) & mask; x ^= x >> sh;
  x = (x + k2) & mask;
  x = (x * 0xC2B2AE35u) & mask; x ^= x >> sh;
  return x;
}
static void robust_offsets(const swimoracle_t* o, uint32_t t, uint32_t* out) {
  uint32_t M = o->N - 1, P = o->P ? o->P : 1, R = (M + P - 1) / P, r = t / R, u = t % R;
  uint32_t rk = tick_key(o->cfg.seed, r);
  uint32_t k1 = hash_h(rk, 0x524F4255u, 1, 0), k2 = hash_h(rk, 0x524F4255u, 2, 0), bits = ceil_log2(M);
  for (uint32_t p = 0; p < o->P; p++) {
    uint64_t k = (uint64_t)u * P + p;
    if (k >= M) { out[p] = 0; continue; }
    uint32_t x = (uint32_t)k;
    if (bits) do x = perm_bits(x, k1, k2, bits); while (x >= M);
    out[p] = 1 + x;
  }
}

/* one period of failureDetector for member i (src/Core.hs:236-240; D14) */
static void failure_detector(octx_t* c, uint32_t i) {
  swimoracle_t* o = c->o;
  if (o->cfg.target_scheme == SWIMSIM_TARGETS_ROBUST) {
    uint32_t off[16], n = 0;
    robust_offsets(o, (uint32_t)o->tick, off);
    for (uint32_t p = 0; p < o->P; p++) if (off[p] && is_alive_in_view(o, i, (i + off[p]) % o->N)) n++;
    o->nsent[i] = (uint8_t)n;
    for (uint32_t p = 0; p < o->P; p++) {
      uint32_t j = (i + off[p]) % o->N;
      if (off[p] && is_alive_in_view(o, i, j)) probe_node(c, i, p, j);   /* probe index = rotation index */
    }
    return;
  }
  uint32_t ms[256];
  uint32_t n = k_random_members(o, i, o->P, NULL, 0, P_SELECT, 0, ms);   /* :239 */
  o->nsent[i] = (uint8_t)n;
  for (uint32_t p = 0; p < n; p++) probe_node(c, i, p, ms[p]);           /* mapM_ :240 */
}

/* ------------------------------------------------------------------------- */
/* end of tick: timers, state rules, piggyback queue                          */
/* ------------------------------------------------------------------------- */
static void timer_push(swimoracle_t* o, uint32_t i, uint32_t s, uint32_t deadline) {
  otimerq_t* q = &o->timers[i];
  if (q->n == q->cap) {
    uint32_t nc = q->cap ? q->cap * 2 : 4;
    otimer_t* nv = (otimer_t*)malloc(nc * sizeof *nv);
    for (uint32_t x = 0; x < q->n; x++) nv[x] = q->v[(q->head + x) % q->cap];
    free(q->v); q->v = nv; q->head = 0; q->cap = nc;
  }
  q->v[(q->head + q->n) % q->cap].subject = s;
  q->v[(q->head + q->n) % q->cap].deadline = deadline;
  q->n++;
}

/* priority of a rumour in the piggyback queue: fewest transmissions first, then
 * lowest subject id (total order => deterministic; H3, D5) */
static inline int rumor_better(const orumor_t* a, const orumor_t* b) {
  if (a->tx != b->tx) return a->tx > b->tx;
  return a->subject < b->subject;
}

/* insert-or-replace into the candidate set, keeping the PB_SLOTS best */
static void cand_insert(opb_t* c, uint32_t subject, uint32_t key, uint8_t tx) {
  orumor_t r; r.subject = subject; r.key = key; r.tx = tx;
  for (int s = 0; s < c->n; s++)
    if (c->r[s].subject == subject) { c->r[s] = r; return; }   /* newer rumour supersedes */
  if (c->n < PB_SLOTS) { c->r[c->n++] = r; return; }
  int worst = 0;
  for (int s = 1; s < c->n; s++) if (rumor_better(&c->r[worst], &c->r[s])) worst = s;
  if (rumor_better(&r, &c->r[worst])) c->r[worst] = r;
}

/* events are collected per worker and appended to the handle's list after the tick (events_fold) */
static void event_add(octx_t* c, uint32_t t, uint32_t i, uint32_t s, uint32_t key, uint8_t cause) {
  if (!(c->o->cfg.event_mask & (1u << cause))) return;
  if (c->nevents == c->events_cap) {
    c->events_cap = c->events_cap ? c->events_cap * 2 : 1024;
    c->events = (swimsim_event_t*)realloc(c->events, c->events_cap * sizeof *c->events);
  }
  swimsim_event_t* e = &c->events[c->nevents++];
  memset(e, 0, sizeof *e);
  e->tick = t; e->observer = i; e->subject = s; e->incarnation = key_inc(key);
  e->state = (uint8_t)key_state(key); e->cause = cause;
}

/* The state rule.  Reference: suspectOrDeadNode' (src/Core.hs:142-187) and the
 * unwritten aliveNode (src/Core.hs:197-218, D6).  Oracle: commutative merge
 * `entry := max(entry, (incarnation,state))` with Alive < Suspect < Dead (H3, D13).
 *   - unknown subject        : reference ignores (:147-148); here every member is
 *                              known from the start (bootstrap assumption).
 *   - i < inc  => ignore      : same (:151).
 *   - about self => refute    : same (:155-166), with D10's counter fix.
 * Returns 1 if i's entry for s changed. */
static int accept_key(octx_t* c, uint32_t i, uint32_t s, uint32_t key, uint8_t cause,
                      uint32_t* refute_inc, opb_t* cand, uint32_t self_inc_start) {
  swimoracle_t* o = c->o;
  uint32_t t = (uint32_t)o->tick;
  if (s == i) {
    /* `name == memberName storeSelf` -> refute (:155); old incarnations ignored (:151) */
    if (key_state(key) != SWIMSIM_ALIVE && key_inc(key) >= self_inc_start)
      if (*refute_inc == NONE32 || key_inc(key) > *refute_inc) *refute_inc = key_inc(key);
    return 0;
  }
  const oentry_t cur = view_get(o, i, s);
  if (o->literal_rule) {                           /* oracle-only mode: the literal rule (D13) */
    const uint32_t mrg = key > cur.key ? key : cur.key;
    const uint32_t lit = key_state(key) == SWIMSIM_ALIVE ? mrg          /* aliveNode is unwritten (D6) */
                                                         : swimoracle_reference_rule(cur.key, key);
    if (lit != mrg) __atomic_fetch_add(&o->d13_hits, 1, __ATOMIC_RELAXED);
    if (lit == cur.key) return 0;
    key = lit;
  } else if (key <= cur.key) return 0;             /* old incarnation / weaker state: ignore */
  oentry_t* e = view_ref(o, i, s);
  if (!e) return 0;
  c->counters[SWIMSIM_CTR_EVDIGEST] += ev_weight(t, i, s) * (uint64_t)(key - cur.key);
  /* every worker writes the same tick into the same few words: look first, so that the line stays shared */
  if (__atomic_load_n(&o->last_change[s], __ATOMIC_RELAXED) != t) __atomic_store_n(&o->last_change[s], t, __ATOMIC_RELAXED);
  if (e->since1 != t + 1) c->counters[SWIMSIM_CTR_CHANGES]++;
  e->key = key; e->since1 = t + 1;                 /* memberLastChange = now (:176) */
  if (cause == SWIMSIM_CAUSE_TIMER) {              /* `deadNode` after the timeout; a false positive if s is up all the same */
    c->counters[SWIMSIM_CTR_TIMERS_FIRED]++;
    if (o->up[s]) c->counters[SWIMSIM_CTR_FALSE_DEADS]++;
  }
  if (key_state(key) == SWIMSIM_SUSPECT) timer_push(o, i, s, t + o->S);   /* D4 */
  cand_insert(cand, s, key, (uint8_t)o->L);        /* `Just msg` -> Broadcast -> enqueue (D5) */
  event_add(c, t, i, s, key, cause);
  return 1;
}

static int pend_cmp_subject_key(const void* a, const void* b) {
  const opend_t* x = (const opend_t*)a; const opend_t* y = (const opend_t*)b;
  if (x->subject != y->subject) return x->subject < y->subject ? -1 : 1;
  return x->key < y->key ? -1 : x->key > y->key;
}

static void end_of_tick(octx_t* c, uint32_t i, const opend_t* pend, uint32_t npend,
                        const ofail_t* fails, uint32_t nfails) {
  swimoracle_t* o = c->o;
  uint32_t t = (uint32_t)o->tick;
  const opb_t* old = &o->pb[i];
  uint32_t self_inc_start = o->self_inc[i];
  uint32_t refute = NONE32;
  /* age the queue: every ping sent this tick carried every slot (D5); a period without a single
   * ping still costs one transmission, so that a rumour is retired after at most L periods */
  opb_t cand; cand.n = 0;
  const uint32_t age = o->nsent[i] ? o->nsent[i] : 1u;
  for (int s = 0; s < old->n; s++)
    if (old->r[s].tx > age) { cand.r[cand.n] = old->r[s]; cand.r[cand.n].tx = (uint8_t)(old->r[s].tx - age); cand.n++; }
  /* phase 1: suspicion timers (the FIXME at src/Core.hs:141; D4), evaluated on the
   * start-of-tick view */
  otimerq_t* q = &o->timers[i];
  uint32_t tprop[64]; uint32_t tkey[64]; uint32_t ntp = 0;
  while (q->n && q->v[q->head].deadline <= t) {
    otimer_t tm = q->v[q->head]; q->head = (q->head + 1) % q->cap; q->n--;
    oentry_t e = view_get(o, i, tm.subject);
    if (key_state(e.key) == SWIMSIM_SUSPECT && e.since1 - 1 + o->S == tm.deadline) {
      if (ntp == 64) { /* flush */
        for (uint32_t x = 0; x < ntp; x++) accept_key(c, i, tprop[x], tkey[x], SWIMSIM_CAUSE_TIMER, &refute, &cand, self_inc_start);
        ntp = 0;
      }
      tprop[ntp] = tm.subject; tkey[ntp] = key_make(key_inc(e.key), SWIMSIM_DEAD); ntp++;
    }
  }
  for (uint32_t x = 0; x < ntp; x++) accept_key(c, i, tprop[x], tkey[x], SWIMSIM_CAUSE_TIMER, &refute, &cand, self_inc_start);
  /* phase 2: own probes that ended without any ack: suspectNode (src/Core.hs:253) with
   * the incarnation of the start-of-tick view entry (`memberIncarnation m`) */
  for (uint32_t f = 0; f < nfails; f++) {
    uint32_t j = fails[f].j;
    uint32_t key = key_make(key_inc(view_get(o, i, j).key), SWIMSIM_SUSPECT);
    accept_key(c, i, j, key, SWIMSIM_CAUSE_PROBE, &refute, &cand, self_inc_start);
  }
  /* phase 3: rumours received this tick (any order: the merge is commutative) */
  if (o->literal_rule && npend > 1) {
    /* the literal rule is not commutative: its order is CANONICAL (include/swimsim.h "Strict reference rules") -- by (subject,
     * incarnation<<2|state) ascending, each rumour applied to what the ones before it left */
    opend_t* srt = (opend_t*)malloc(npend * sizeof *srt);
    memcpy(srt, pend, npend * sizeof *srt);
    qsort(srt, npend, sizeof *srt, pend_cmp_subject_key);
    for (uint32_t x = 0; x < npend; x++)
      accept_key(c, i, srt[x].subject, srt[x].key, SWIMSIM_CAUSE_GOSSIP, &refute, &cand, self_inc_start);
    free(srt);
  } else if (o->shuffle_seed && npend > 1) {
    uint32_t* perm = (uint32_t*)malloc(npend * sizeof *perm);
    for (uint32_t x = 0; x < npend; x++) perm[x] = x;
    for (uint32_t x = npend - 1; x > 0; x--) {
      uint32_t r = (uint32_t)(mix64(o->shuffle_seed + ((uint64_t)t << 32) + (uint64_t)i * 0x9E3779B97F4A7C15ull + x) % (x + 1));
      uint32_t tmp = perm[x]; perm[x] = perm[r]; perm[r] = tmp;
    }
    for (uint32_t x = 0; x < npend; x++)
      accept_key(c, i, pend[perm[x]].subject, pend[perm[x]].key, SWIMSIM_CAUSE_GOSSIP, &refute, &cand, self_inc_start);
    free(perm);
  } else {
    for (uint32_t x = 0; x < npend; x++)
      accept_key(c, i, pend[x].subject, pend[x].key, SWIMSIM_CAUSE_GOSSIP, &refute, &cand, self_inc_start);
  }
  /* refutation: bump own incarnation past the rumour's (src/Core.hs:155-166; D10) */
  if (refute != NONE32) {
    uint32_t ni = refute + 1;
    if (ni > INC_MAX) { fail(o, SWIMSIM_ERR_CAPACITY, "incarnation overflow"); return; }
    o->self_inc[i] = ni;
    c->counters[SWIMSIM_CTR_EVDIGEST] += h4(TAG_INC, ((uint64_t)t << 32) | i, ni, 0);
    c->counters[SWIMSIM_CTR_REFUTES]++;
    cand_insert(&cand, i, key_make(ni, SWIMSIM_ALIVE), (uint8_t)o->L);   /* Just Alive{..} :163 */
    __atomic_store_n(&o->last_change[i], t, __ATOMIC_RELAXED);
    event_add(c, t, i, i, key_make(ni, SWIMSIM_ALIVE), SWIMSIM_CAUSE_REFUTE);
  }
  if (old->n > 0 || cand.n > 0) c->counters[SWIMSIM_CTR_PB_WRITES]++;
  o->pb[i] = cand;
}

/* ------------------------------------------------------------------------- */
/* end of tick with bounded member maps (view_cap; include/swimsim.h)          */
/* ------------------------------------------------------------------------- */
/* The same rules -- suspicion timers (D4), own failed probes (src/Core.hs:253), received rumours through
 * suspectOrDeadNode' / aliveNode as the max-merge (src/Core.hs:142-218; H3, D6, D13), refutation (:155-166; D10), the
 * piggyback queue (D5) -- stated over SETS, because a capacity makes "apply one proposal after the other" depend on the
 * order: proposals -> the largest per subject -> entries that grow are CHANGED (lastChange = t) -> the C entries with the
 * largest (lastChange, rank) stay, the rest is evicted (back to the default) -> what changed AND stayed is accounted
 * (counters, events, digest, queue).  An entry that appears or changes and is evicted in the same tick has no effect. */
typedef struct { uint32_t subject, key, prio; } oprop_t;          /* prio: 2 timer, 1 own probe, 0 gossip */
static int prop_cmp(const void* a, const void* b) {
  const oprop_t* x = (const oprop_t*)a; const oprop_t* y = (const oprop_t*)b;
  if (x->subject != y->subject) return x->subject < y->subject ? -1 : 1;
  if (x->key != y->key) return x->key > y->key ? -1 : 1;          /* the largest key first ... */
  return x->prio > y->prio ? -1 : x->prio < y->prio;              /* ... stated by the earliest phase */
}
typedef struct { osent_t e; uint32_t k0, rank; uint8_t changed, cause, due, was; } ocand_t;
/* who stays when the map is over capacity: the most recent lastChange first; among entries of one tick the order is a
 * keyed permutation of the subject ids -- rank = mix32(subject ^ H(t, i)): mix32 is a bijection, so no two subjects tie,
 * and no member is forgotten first by everybody all the time (an order by subject id would be that) */
static int evict_cmp(const void* a, const void* b) {              /* best first */
  const ocand_t* x = (const ocand_t*)a; const ocand_t* y = (const ocand_t*)b;
  if (x->e.since1 != y->e.since1) return x->e.since1 > y->e.since1 ? -1 : 1;
  return x->rank > y->rank ? -1 : x->rank < y->rank;
}
static int cand_subject_cmp(const void* a, const void* b) {
  uint32_t x = ((const ocand_t*)a)->e.subject, y = ((const ocand_t*)b)->e.subject;
  return x < y ? -1 : x > y;
}

static void end_of_tick_sparse(octx_t* c, uint32_t i, const opend_t* pend, uint32_t npend,
                               const ofail_t* fails, uint32_t nfails) {
  swimoracle_t* o = c->o;
  const uint32_t t = (uint32_t)o->tick;
  otab_t* tb = &o->tab[i];
  const opb_t* old = &o->pb[i];
  const uint32_t self_inc_start = o->self_inc[i];
  uint32_t refute = NONE32;
  oprop_t* props = (oprop_t*)malloc(((size_t)tb->n + nfails + npend + 1) * sizeof *props);
  size_t np = 0;
  /* suspicion deadlines that are due (the FIXME at src/Core.hs:141; D4): Suspect since t' with t' + S <= t */
  for (uint32_t x = 0; x < tb->n; x++) {
    const osent_t* e = &tb->v[x];
    if (key_state(e->key) == SWIMSIM_SUSPECT && e->since1 - 1 + o->S <= t) {
      props[np].subject = e->subject; props[np].key = key_make(key_inc(e->key), SWIMSIM_DEAD); props[np].prio = 2; np++;
    }
  }
  /* own probes that ended without any ack: suspectNode (src/Core.hs:253) at the incarnation the map holds */
  for (uint32_t f = 0; f < nfails; f++) {
    const uint32_t j = fails[f].j;
    props[np].subject = j; props[np].key = key_make(key_inc(view_get(o, i, j).key), SWIMSIM_SUSPECT); props[np].prio = 1; np++;
  }
  /* rumours received this tick; about self -> refute (src/Core.hs:155-166), old incarnations ignored (:151) */
  for (uint32_t x = 0; x < npend; x++) {
    if (pend[x].subject == i) {
      if (key_state(pend[x].key) != SWIMSIM_ALIVE && key_inc(pend[x].key) >= self_inc_start)
        if (refute == NONE32 || key_inc(pend[x].key) > refute) refute = key_inc(pend[x].key);
      continue;
    }
    props[np].subject = pend[x].subject; props[np].key = pend[x].key; props[np].prio = 0; np++;
  }
  qsort(props, np, sizeof *props, prop_cmp);
  /* the map after the merge: every old entry, every subject whose best proposal beats the entry / the default */
  ocand_t* cand = (ocand_t*)malloc(((size_t)tb->n + np + 1) * sizeof *cand);
  size_t nc = 0, xp = 0; uint32_t xt = 0;
  while (xt < tb->n || xp < np) {
    const uint32_t st = xt < tb->n ? tb->v[xt].subject : NONE32, sp = xp < np ? props[xp].subject : NONE32;
    const uint32_t sj = st < sp ? st : sp;
    ocand_t cd; memset(&cd, 0, sizeof cd);
    cd.e.subject = sj;
    if (st == sj) { cd.e = tb->v[xt]; cd.k0 = cd.e.key; cd.was = 1; cd.due = key_state(cd.e.key) == SWIMSIM_SUSPECT && cd.e.since1 - 1 + o->S <= t; xt++; }
    else { cd.e.key = o->base[sj]; cd.k0 = cd.e.key; cd.e.since1 = 0; }
    if (sp == sj) {
      const oprop_t* best = &props[xp];                /* sorted: the first of its subject is the winner */
      if (best->key > cd.e.key) {
        cd.e.key = best->key; cd.e.since1 = t + 1; cd.changed = 1;   /* memberLastChange = now (:176) */
        cd.cause = best->prio == 2 ? SWIMSIM_CAUSE_TIMER : best->prio == 1 ? SWIMSIM_CAUSE_PROBE : SWIMSIM_CAUSE_GOSSIP;
      }
      while (xp < np && props[xp].subject == sj) xp++;
    }
    if (st == sj || cd.changed) cand[nc++] = cd;       /* a proposal that loses against the default leaves nothing */
  }
  /* the capacity: the C entries with the largest (lastChange, subject) stay */
  if (nc > o->C) {
    const uint32_t mk = mix32(o->tk ^ i);
    for (size_t x = 0; x < nc; x++) cand[x].rank = mix32(cand[x].e.subject ^ mk);
    qsort(cand, nc, sizeof *cand, evict_cmp);
    for (size_t x = o->C; x < nc; x++) c->counters[SWIMSIM_CTR_EVICTED] += cand[x].was;   /* entries of the start of the tick that leave */
    nc = o->C;
    qsort(cand, nc, sizeof *cand, cand_subject_cmp);
  }
  /* the queue: aged survivors, then what changed and stayed (D5) */
  opb_t q; q.n = 0;
  const uint32_t age = o->nsent[i] ? o->nsent[i] : 1u;
  for (int sl = 0; sl < old->n; sl++)
    if (old->r[sl].tx > age) { q.r[q.n] = old->r[sl]; q.r[q.n].tx = (uint8_t)(old->r[sl].tx - age); q.n++; }
  for (size_t x = 0; x < nc; x++) {
    tb->v[x] = cand[x].e;
    if (!cand[x].changed) continue;
    const uint32_t sj = cand[x].e.subject, key = cand[x].e.key;
    c->counters[SWIMSIM_CTR_EVDIGEST] += ev_weight(t, i, sj) * (uint64_t)(key - cand[x].k0);
    c->counters[SWIMSIM_CTR_CHANGES]++;
    if (cand[x].due) {                                 /* `deadNode` after the timeout; a false positive if the subject is up */
      c->counters[SWIMSIM_CTR_TIMERS_FIRED]++;
      if (o->up[sj]) c->counters[SWIMSIM_CTR_FALSE_DEADS]++;
    }
    cand_insert(&q, sj, key, (uint8_t)o->L);
    event_add(c, t, i, sj, key, cand[x].cause);
  }
  tb->n = (uint32_t)nc;
  free(cand); free(props);
  if (refute != NONE32) {                              /* src/Core.hs:155-166; D10 */
    uint32_t ni = refute + 1;
    if (ni > INC_MAX) { fail(o, SWIMSIM_ERR_CAPACITY, "incarnation overflow"); return; }
    o->self_inc[i] = ni;
    c->counters[SWIMSIM_CTR_EVDIGEST] += h4(TAG_INC, ((uint64_t)t << 32) | i, ni, 0);
    c->counters[SWIMSIM_CTR_REFUTES]++;
    cand_insert(&q, i, key_make(ni, SWIMSIM_ALIVE), (uint8_t)o->L);
    event_add(c, t, i, i, key_make(ni, SWIMSIM_ALIVE), SWIMSIM_CAUSE_REFUTE);
  }
  if (old->n > 0 || q.n > 0) c->counters[SWIMSIM_CTR_PB_WRITES]++;
  o->pb[i] = q;
}

/* ------------------------------------------------------------------------- */
/* faults                                                                     */
/* ------------------------------------------------------------------------- */
static int fault_cmp(const void* a, const void* b) {
  const ofault_t* x = (const ofault_t*)a; const ofault_t* y = (const ofault_t*)b;
  if (x->tick != y->tick) return x->tick < y->tick ? -1 : 1;
  return x->order < y->order ? -1 : x->order > y->order;
}

/* State pulls (include/swimsim.h): member m merges the member map of a host -- a member that is up, has
 * no change scheduled in this tick (faults[0..nf) are the tick's changes) and, with pull_ticks = T on, is not one of the
 * tick's periodic pullers (nobody reads a map that is being written).  purpose = P_JOIN: m just came up (`joinHosts`,
 * src/Types.hs:47); P_PULL: m's periodic pull (the commented-out PushPullMsg, src/Types.hs:165,177). */
static uint32_t pull_host_of(swimoracle_t* o, uint32_t t, uint32_t m, size_t nf, uint32_t purpose) {
  uint32_t tk = tick_key(o->cfg.seed, t), h = NONE32;
  const uint32_t T = o->cfg.pull_ticks;
  for (uint32_t a = 0; a < SEL_ATTEMPTS && h == NONE32; a++) {
    uint32_t c = (uint32_t)(((uint64_t)hash_h(tk, m, (purpose << 24) | a, 0) * o->N) >> 32);
    int busy = (c == m) || !o->up[c] || (T && c % T == t % T);
    for (size_t x = 0; x < nf && !busy; x++) busy = o->faults[x].member == c;
    if (!busy) h = c;
  }
  return h;
}
/* m merges h's member map (`from` = h, a pull) -- or, push half of a push-pull (include/swimsim.h "Periodic state pull"): called with
 * the roles swapped, the host merges its puller's map.  The sender counts as Alive at its own incarnation. */
static void state_merge_from(swimoracle_t* o, uint32_t t, uint32_t m, uint32_t h) {
  for (uint32_t sl = 0; sl < o->nslots; sl++) {
    if (o->free_at[sl] != NONE32) continue;
    uint32_t s = o->subject_of[sl];
    if (s == m) continue;
    uint32_t kh = s == h ? key_make(o->self_inc[h], SWIMSIM_ALIVE) : view_get(o, h, s).key;
    oentry_t cur = view_get(o, m, s);
    if (kh <= cur.key) continue;
    oentry_t* e = view_ref(o, m, s);
    o->counters[SWIMSIM_CTR_EVDIGEST] += ev_weight(t, m, s) * (uint64_t)(kh - cur.key);
    if (e->since1 != t + 1) o->counters[SWIMSIM_CTR_CHANGES]++;   /* (a host raised by two pullers in one tick: one changed entry) */
    e->key = kh; e->since1 = t + 1;
    o->last_change[s] = t;
    if (key_state(kh) == SWIMSIM_SUSPECT) timer_push(o, m, s, t + o->S);
  }
}
static void state_pull(swimoracle_t* o, uint32_t t, uint32_t m, size_t nf, uint32_t purpose) {
  const uint32_t h = pull_host_of(o, t, m, nf, purpose);
  if (h != NONE32) state_merge_from(o, t, m, h);
}

static void apply_faults(swimoracle_t* o, uint32_t t) {
  size_t k = 0, nf = 0;
  while (nf < o->nfaults && o->faults[nf].tick <= t) nf++;
  while (k < o->nfaults && o->faults[k].tick <= t) {
    ofault_t f = o->faults[k++];
    uint32_t m = f.member;
    if (f.up == o->up[m]) continue;
    o->up[m] = f.up;
    if (!f.up) {
      o->crash_tick[m] = t; o->first_suspect[m] = NONE32; o->pb[m].n = 0;   /* the process's queue is lost ... */
      /* ... and its inbox: a message from outside the simulation reaches a member that STAYS up through the tick's
       * scheduled changes (include/swimsim.h) -- down and up again in one tick is a new process that was not listening */
      for (size_t x = 0; x < o->ninj; x++) if (o->inj[x].dst == m) o->inj[x].dst = NONE32;
    }
    else {
      /* (re)join: new incarnation + announce Alive (memberlist-style; the reference's
       * joinHosts is dead config, src/Util.hs:46) */
      uint32_t ni = o->self_inc[m] + 1;
      if (ni > INC_MAX) { fail(o, SWIMSIM_ERR_CAPACITY, "incarnation overflow"); return; }
      o->self_inc[m] = ni;
      o->counters[SWIMSIM_CTR_EVDIGEST] += h4(TAG_INC, ((uint64_t)t << 32) | m, ni, 0);
      cand_insert(&o->pb[m], m, key_make(ni, SWIMSIM_ALIVE), (uint8_t)o->L);
      /* the announcement is a rumour about m: m has a view column from now on, whether or not anybody ever stores
       * an entry in it (a member that goes down again before it is heard of: the column settles empty; counter 15,
       * max_subjects and base_since see it) */
      if (!o->C && !view_ref(o, m, m)) return;
      o->last_change[m] = t;
      event_add(&o->ctx[0], t, m, m, key_make(ni, SWIMSIM_ALIVE), SWIMSIM_CAUSE_JOIN);
      o->first_suspect[m] = NONE32;
      if (o->cfg.join_pull) state_pull(o, t, m, nf, P_JOIN);
    }
  }
  /* the periodic pulls of the tick: members i = t mod T (mod T) that are up and have no change in this tick */
  if (o->cfg.pull_ticks) {
    const uint32_t T = o->cfg.pull_ticks;
    for (uint32_t i = t % T; i < o->N; i += T) {
      int busy = !o->up[i];
      for (size_t x = 0; x < nf && !busy; x++) busy = o->faults[x].member == i;
      if (!busy) state_pull(o, t, i, nf, P_PULL);
    }
    /* push_pull: after ALL pulls, every puller's host merges the puller's map (hosts are never pullers: the pulls above read maps
     * nobody wrote, the pushes write maps nobody reads; several pullers of one host: max over all, any order) */
    if (o->cfg.push_pull)
      for (uint32_t i = t % T; i < o->N; i += T) {
        int busy = !o->up[i];
        for (size_t x = 0; x < nf && !busy; x++) busy = o->faults[x].member == i;
        if (busy) continue;
        const uint32_t h = pull_host_of(o, t, i, nf, P_PULL);
        if (h != NONE32) state_merge_from(o, t, h, i);
      }
  }
  if (k) { memmove(o->faults, o->faults + k, (o->nfaults - k) * sizeof *o->faults); o->nfaults -= k; }
}

int swimoracle_schedule_fault(swimoracle_t* o, uint64_t tick, uint32_t member, uint8_t up) {
  if (!o) return SWIMSIM_ERR_INVALID;
  if (member >= o->N || up > 1) return fail(o, SWIMSIM_ERR_INVALID, "schedule_fault: bad member/up");
  if (tick < o->tick || tick >= 0xFFFFFFFEull) return fail(o, SWIMSIM_ERR_INVALID, "schedule_fault: tick in the past");
  if (o->nfaults == o->faults_cap) {
    o->faults_cap = o->faults_cap ? o->faults_cap * 2 : 64;
    o->faults = (ofault_t*)realloc(o->faults, o->faults_cap * sizeof *o->faults);
  }
  ofault_t f; f.tick = (uint32_t)tick; f.member = member; f.up = up; f.order = o->fault_order++;
  o->faults[o->nfaults++] = f;
  qsort(o->faults, o->nfaults, sizeof *o->faults, fault_cmp);
  return SWIMSIM_OK;
}

/* ------------------------------------------------------------------------- */
/* the tick                                                                   */
/* ------------------------------------------------------------------------- */
/* phase A for one worker: one period of failureDetector for each of its up members */
static void phase_probe(octx_t* c) {
  swimoracle_t* o = c->o;
  c->nfails = 0;
  for (uint32_t g = 0; g < o->nworkers; g++) c->out[g].n = 0;
  /* messages from outside the simulation reach their members like any datagram of the tick (src/Core.hs:110-117) */
  if (c->w == 0) for (size_t x = 0; x < o->ninj; x++)
    if (o->inj[x].dst != NONE32 && o->up[o->inj[x].dst]) pend_add(c, o->inj[x].dst, o->inj[x].subject, o->inj[x].key);
  for (uint32_t i = c->lo; i < c->hi; i++) {
    o->nsent[i] = 0;
    if (o->up[i]) { failure_detector(c, i); c->counters[SWIMSIM_CTR_ACTIVE_MEMBERS]++; }
  }
}

/* phase B for one worker: the rumours delivered to its members (from every worker's lists), bucketed
 * by receiver (stable: arrival order kept), then every up member's end of tick */
static void phase_merge(octx_t* c) {
  swimoracle_t* o = c->o;
  const uint32_t n = c->hi - c->lo;
  size_t total = 0;
  for (uint32_t g = 0; g < o->nworkers; g++) total += o->ctx[g].out[c->w].n;
  if (total > c->sorted_cap) {
    c->sorted_cap = total * 2;
    c->sorted = (opend_t*)realloc(c->sorted, c->sorted_cap * sizeof *c->sorted);
  }
  memset(c->off, 0, ((size_t)n + 1) * sizeof *c->off);
  for (uint32_t g = 0; g < o->nworkers; g++) {
    const opendv_t* q = &o->ctx[g].out[c->w];
    for (size_t x = 0; x < q->n; x++) c->off[q->v[x].dst - c->lo + 1]++;
  }
  for (uint32_t i = 0; i < n; i++) c->off[i + 1] += c->off[i];
  {
    uint32_t* cur = (uint32_t*)malloc(((size_t)n + 1) * sizeof *cur);
    memcpy(cur, c->off, (size_t)n * sizeof *cur);
    for (uint32_t g = 0; g < o->nworkers; g++) {
      const opendv_t* q = &o->ctx[g].out[c->w];
      for (size_t x = 0; x < q->n; x++) c->sorted[cur[q->v[x].dst - c->lo]++] = q->v[x];
    }
    free(cur);
  }
  size_t fc = 0;
  for (uint32_t i = c->lo; i < c->hi; i++) {
    size_t f0 = fc;
    while (fc < c->nfails && c->fails[fc].i == i) fc++;
    if (!o->up[i]) continue;
    if (o->C) end_of_tick_sparse(c, i, c->sorted + c->off[i - c->lo], c->off[i - c->lo + 1] - c->off[i - c->lo],
                                 c->fails + f0, (uint32_t)(fc - f0));
    else end_of_tick(c, i, c->sorted + c->off[i - c->lo], c->off[i - c->lo + 1] - c->off[i - c->lo],
                     c->fails + f0, (uint32_t)(fc - f0));
    if (__atomic_load_n(&o->poisoned, __ATOMIC_RELAXED)) return;
  }
}

static void run_phase(octx_t* c, int phase) { if (phase == 0) phase_probe(c); else phase_merge(c); }

static void* worker_main(void* arg) {
  octx_t* c = (octx_t*)arg;
  swimoracle_t* o = c->o;
  for (;;) {
    pthread_barrier_wait(&o->bar);                /* phase published */
    if (o->quit) return NULL;
    run_phase(c, o->phase);
    pthread_barrier_wait(&o->bar);                /* phase done */
  }
}

static void parallel_phase(swimoracle_t* o, int phase) {
  if (o->nworkers == 1) { run_phase(&o->ctx[0], phase); return; }
  o->phase = phase;
  pthread_barrier_wait(&o->bar);
  run_phase(&o->ctx[0], phase);                   /* the calling thread is worker 0 */
  pthread_barrier_wait(&o->bar);
}

/* counters add up; events are appended in worker order (drain sorts them) */
static void fold_workers(swimoracle_t* o) {
  for (uint32_t g = 0; g < o->nworkers; g++) {
    octx_t* c = &o->ctx[g];
    for (int k = 0; k < SWIMSIM_CTR_COUNT; k++) { o->counters[k] += c->counters[k]; c->counters[k] = 0; }
    for (size_t x = 0; x < c->nevents; x++) {
      if (o->nevents >= o->cfg.event_cap) { o->counters[SWIMSIM_CTR_EVENTS_DROPPED]++; continue; }
      if (o->nevents == o->events_cap_alloc) {
        o->events_cap_alloc = o->events_cap_alloc ? o->events_cap_alloc * 2 : 1024;
        o->events = (swimsim_event_t*)realloc(o->events, o->events_cap_alloc * sizeof *o->events);
      }
      o->events[o->nevents++] = c->events[x];
    }
    c->nevents = 0;
  }
}

/* Settling at the end of tick t (include/swimsim.h): `removeDeadNodes` (src/Core.hs:65-67)
 * + the anti-entropy the reference leaves commented out (PushPullMsg, src/Types.hs:165,177). */
static void settle(swimoracle_t* o, uint32_t t) {
  if (!o->G) return;
  for (uint32_t sl = 0; sl < o->nslots; sl++) {
    if (o->free_at[sl] != NONE32) continue;
    const uint32_t s = o->subject_of[sl];
    const uint32_t lc = o->last_change[s];
    if (lc != NONE32 && t - lc < o->G) continue;       /* lc == NONE32: column made by set_view only */
    uint32_t k = 0;
    oentry_t* col = o->cols[sl];
    for (uint32_t i = 0; i < o->N; i++) if (o->up[i] && i != s && col[i].key > k) k = col[i].key;
    if (key_state(k) == SWIMSIM_SUSPECT) continue;     /* somebody's timer is still running */
    if (o->N <= 4096)                                  /* the invariant G >= S + L + 2 buys (checked where cheap) */
      for (uint32_t i = 0; i < o->N; i++)
        for (int q = 0; q < o->pb[i].n; q++)
          if (o->pb[i].r[q].subject == s) { fail(o, SWIMSIM_ERR_STATE, "settle: a queue still holds the subject"); return; }
    if (k > o->base[s]) o->base[s] = k;
    o->base_since[s] = t;
    memset(col, 0, (size_t)o->N * sizeof *col);
    o->slot_of[s] = 0;
    o->free_at[sl] = t + 2;
    o->nlive--;
    o->counters[SWIMSIM_CTR_SETTLED]++;
  }
}

static int one_tick(swimoracle_t* o) {
  uint32_t t = (uint32_t)o->tick;
  /* A message from outside the simulation (swimsim_inject_rumor) reaches a member that is up when the tick starts and
   * stays up through the tick's scheduled changes.  A rumour somebody states opens the subject's column even if the
   * receiver ignores it -- at the start of the tick, so that a column nobody holds anything in settles
   * at the end of it. */
  for (size_t x = 0; x < o->ninj; x++) {
    if (!o->up[o->inj[x].dst]) o->inj[x].dst = NONE32;            /* nobody listening */
    else if (!view_ref(o, o->inj[x].dst, o->inj[x].subject)) return SWIMSIM_ERR_CAPACITY;
  }
  apply_faults(o, t);
  fold_workers(o);                                /* JOIN events */
  if (o->poisoned) return SWIMSIM_ERR_CAPACITY;
  o->tk = tick_key(o->cfg.seed, t);
  parallel_phase(o, 0);
  parallel_phase(o, 1);
  fold_workers(o);
  if (o->poisoned) return SWIMSIM_ERR_CAPACITY;
  settle(o, t);
  if (o->poisoned) return SWIMSIM_ERR_CAPACITY;
  o->ninj = 0;
  o->tick++;
  return SWIMSIM_OK;
}

static void workers_stop(swimoracle_t* o) {
  if (o->nworkers > 1) {
    o->quit = 1;
    pthread_barrier_wait(&o->bar);
    for (uint32_t g = 1; g < o->nworkers; g++) pthread_join(o->threads[g], NULL);
    pthread_barrier_destroy(&o->bar);
    o->quit = 0;
  }
  free(o->threads); o->threads = NULL;
  for (uint32_t g = 0; g < o->nworkers; g++) {
    octx_t* c = &o->ctx[g];
    for (uint32_t d = 0; c->out && d < o->nworkers; d++) free(c->out[d].v);
    free(c->out); free(c->fails); free(c->events); free(c->sorted); free(c->off);
  }
  free(o->ctx); o->ctx = NULL; o->nworkers = 0;
}

/* (re)partition the members over n workers; n-1 threads are started (the caller is worker 0) */
static int workers_start(swimoracle_t* o, uint32_t n) {
  if (n < 1) n = 1;
  if (n > o->N) n = o->N;
  uint32_t chunk = (o->N + n - 1) / n;
  n = (o->N + chunk - 1) / chunk;
  o->ctx = (octx_t*)aligned_alloc(128, (size_t)n * sizeof *o->ctx);
  if (!o->ctx) return SWIMSIM_ERR_NOMEM;
  memset(o->ctx, 0, (size_t)n * sizeof *o->ctx);
  o->nworkers = n; o->chunk = chunk;
  for (uint32_t g = 0; g < n; g++) {
    octx_t* c = &o->ctx[g];
    c->o = o; c->w = g; c->lo = g * chunk; c->hi = c->lo + chunk > o->N ? o->N : c->lo + chunk;
    c->out = (opendv_t*)calloc(n, sizeof *c->out);
    c->off = (uint32_t*)calloc((size_t)(c->hi - c->lo) + 1, sizeof *c->off);
    if (!c->out || !c->off) return SWIMSIM_ERR_NOMEM;
  }
  if (n > 1) {
    o->threads = (pthread_t*)calloc(n, sizeof *o->threads);
    if (!o->threads || pthread_barrier_init(&o->bar, NULL, n)) return SWIMSIM_ERR_NOMEM;
    for (uint32_t g = 1; g < n; g++)
      if (pthread_create(&o->threads[g], NULL, worker_main, &o->ctx[g])) return SWIMSIM_ERR_NOMEM;
  }
  return SWIMSIM_OK;
}

int swimoracle_set_threads(swimoracle_t* o, uint32_t n) {
  if (!o) return SWIMSIM_ERR_INVALID;
  fold_workers(o);
  workers_stop(o);
  return workers_start(o, n) ? fail(o, SWIMSIM_ERR_NOMEM, "out of memory (workers)") : SWIMSIM_OK;
}

int swimoracle_step(swimoracle_t* o, uint32_t nticks) {
  if (!o) return SWIMSIM_ERR_INVALID;
  if (o->poisoned) return fail(o, SWIMSIM_ERR_STATE, "handle is poisoned by an earlier capacity error");
  for (uint32_t k = 0; k < nticks; k++) {
    int rc = one_tick(o);
    if (rc) return rc;
  }
  return SWIMSIM_OK;
}

int swimoracle_tick(const swimoracle_t* o, uint64_t* tick) {
  if (!o || !tick) return SWIMSIM_ERR_INVALID;
  *tick = o->tick; return SWIMSIM_OK;
}

/* ------------------------------------------------------------------------- */
/* create / destroy                                                           */
/* ------------------------------------------------------------------------- */
static int resolve_config(const swimsim_config_t* in, swimsim_config_t* c, char* err, size_t errn) {
  if (!in) { snprintf(err, errn, "config is NULL"); return SWIMSIM_ERR_INVALID; }
  if (in->struct_size != sizeof *in || in->abi_version != SWIMSIM_ABI_VERSION) {
    snprintf(err, errn, "config struct_size/abi_version mismatch"); return SWIMSIM_ERR_INVALID; }
  *c = *in;
  if (c->n_members < 2 || c->n_members > 0x7FFFFFFFu) { snprintf(err, errn, "n_members must be in [2, 2^31)"); return SWIMSIM_ERR_INVALID; }
  if (c->num_to_gossip < 0) { snprintf(err, errn, "num_to_gossip must be >= 0"); return SWIMSIM_ERR_INVALID; }
  if (c->probes_per_tick == 0) c->probes_per_tick = c->num_to_gossip;
  if (c->indirect_k == 0) c->indirect_k = c->num_to_gossip;
  if (c->probes_per_tick < 0 || c->probes_per_tick > 16 || c->indirect_k < 0 || c->indirect_k > 16) {
    snprintf(err, errn, "probes_per_tick / indirect_k must be in [0,16]"); return SWIMSIM_ERR_INVALID; }
  if (c->loss_ppm > 1000000u) { snprintf(err, errn, "loss_ppm must be <= 1000000"); return SWIMSIM_ERR_INVALID; }
  if (c->suspicion_ticks == 0) c->suspicion_ticks = 3 * ceil_log2(c->n_members);
  if (c->suspicion_ticks == 0) c->suspicion_ticks = 1;
  if (c->retransmit_mult == 0) c->retransmit_mult = 3;
  if ((uint64_t)c->retransmit_mult * ceil_log2((uint64_t)c->n_members + 1) > 255) {
    snprintf(err, errn, "retransmit budget exceeds 255"); return SWIMSIM_ERR_INVALID; }
  if (c->max_subjects == 0) {
    /* enough view columns for the subjects of ~256 periods of false suspicions (all in parts per million,
     * integers only: the oracle and the product must agree on the number), within 32 GB of columns */
    uint64_t q = 1000000u - c->loss_ppm, q2 = q * q / 1000000u, q4 = q2 * q2 / 1000000u;
    uint64_t pf = 1000000u - q2;                                   /* the direct probe fails ... */
    for (int k = 0; k < c->indirect_k; k++) pf = pf * (1000000u - q4) / 1000000u;   /* ... and every proxy chain */
    uint64_t est = pf * (uint64_t)c->probes_per_tick * c->n_members / 1000000u * 512u + 1024u;
    uint64_t mem = 32000000000ull / (8ull * c->n_members);
    if (est > mem) est = mem < 64 ? 64 : mem;
    if (est > 60000u) est = 60000u;
    if (est > c->n_members) est = c->n_members;
    c->max_subjects = (uint32_t)est;
  }
  if (c->max_subjects > 60000u) { snprintf(err, errn, "max_subjects must be <= 60000"); return SWIMSIM_ERR_INVALID; }
  {
    uint32_t gmin = c->suspicion_ticks + c->retransmit_mult * ceil_log2((uint64_t)c->n_members + 1) + 2;
    if (c->gc_ticks == SWIMSIM_GC_AUTO) c->gc_ticks = gmin;
    if (c->gc_ticks && c->gc_ticks < gmin) { snprintf(err, errn, "gc_ticks must be 0, SWIMSIM_GC_AUTO or >= suspicion_ticks + L + 2 = %u", gmin); return SWIMSIM_ERR_INVALID; }
  }
  if (c->event_cap == 0) c->event_cap = 1u << 20;
  if (c->event_mask == 0) c->event_mask = SWIMSIM_EVMASK_DEFAULT;
  if (c->n_shards == 0) c->n_shards = 1;
  if (c->n_shards != 1 || c->shard_index != 0) { snprintf(err, errn, "oracle: sharding is driven from outside (n_shards must be 1)"); return SWIMSIM_ERR_INVALID; }
  if (c->target_scheme > SWIMSIM_TARGETS_ROBUST) { snprintf(err, errn, "unknown target_scheme"); return SWIMSIM_ERR_INVALID; }
  if (c->join_pull > 1) { snprintf(err, errn, "join_pull must be 0 or 1"); return SWIMSIM_ERR_INVALID; }
  if (c->pull_ticks == 1) { snprintf(err, errn, "pull_ticks must be 0 (off) or >= 2"); return SWIMSIM_ERR_INVALID; }
  if (c->push_pull > 1 || (c->push_pull && !c->pull_ticks)) { snprintf(err, errn, "push_pull must be 0 or 1 and needs pull_ticks"); return SWIMSIM_ERR_INVALID; }
  if (c->strict_reference_rules > 1) { snprintf(err, errn, "strict_reference_rules must be 0 or 1"); return SWIMSIM_ERR_INVALID; }
  if (c->strict_reference_rules && c->view_cap) {
    snprintf(err, errn, "strict_reference_rules cannot be combined with view_cap"); return SWIMSIM_ERR_INVALID; }
  if (c->view_cap) {
    if (c->view_cap < SWIMSIM_VIEW_CAP_MIN || c->view_cap > SWIMSIM_VIEW_CAP_MAX) { snprintf(err, errn, "view_cap must be 0 (unbounded) or in [%u, %u]", SWIMSIM_VIEW_CAP_MIN, SWIMSIM_VIEW_CAP_MAX); return SWIMSIM_ERR_INVALID; }
    if (c->gc_ticks || c->join_pull || c->pull_ticks || c->target_scheme != SWIMSIM_TARGETS_RANDOM) {
      snprintf(err, errn, "view_cap (bounded member maps) cannot be combined with gc_ticks, join_pull, pull_ticks or the robust target scheme"); return SWIMSIM_ERR_INVALID; }
  }
  return SWIMSIM_OK;
}

/* makeStore / makeSelf for every member (src/Util.hs:76-101): incarnation 0, seqNo 0,
 * empty gossip queue; bootstrap assumption: everyone knows everyone as Alive@0. */
int swimoracle_create(const swimsim_config_t* cfg, swimoracle_t** out) {
  if (!out) return SWIMSIM_ERR_INVALID;
  *out = NULL;
  swimsim_config_t c;
  int rc = resolve_config(cfg, &c, g_create_err, sizeof g_create_err);
  if (rc) return rc;
  swimoracle_t* o = (swimoracle_t*)calloc(1, sizeof *o);
  if (!o) return fail(NULL, SWIMSIM_ERR_NOMEM, "out of memory");
  o->cfg = c; o->N = c.n_members; o->P = (uint32_t)c.probes_per_tick; o->K = (uint32_t)c.indirect_k;
  o->S = c.suspicion_ticks; o->L = c.retransmit_mult * ceil_log2((uint64_t)c.n_members + 1);
  o->literal_rule = c.strict_reference_rules != 0;   /* include/swimsim.h "Strict reference rules" (D13) */
  {
    uint64_t thr = ((uint64_t)c.loss_ppm << 32) / 1000000ull;
    o->loss_thr = thr > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)thr;
  }
  uint32_t N = o->N;
  o->up = (uint8_t*)malloc(N); o->self_inc = (uint32_t*)calloc(N, 4);
  o->pb = (opb_t*)calloc(N, sizeof(opb_t)); o->timers = (otimerq_t*)calloc(N, sizeof(otimerq_t));
  o->nsent = (uint8_t*)calloc(N, 1); o->slot_of = (uint32_t*)calloc(N, 4);
  o->first_suspect = (uint32_t*)malloc((size_t)N * 4); o->crash_tick = (uint32_t*)malloc((size_t)N * 4);
  /* settled columns wait two ticks before reuse, so a few more than max_subjects may exist */
  o->slots_cap = c.max_subjects + (c.gc_ticks ? c.max_subjects : 0);
  o->cols = (oentry_t**)calloc(o->slots_cap, sizeof *o->cols);
  o->subject_of = (uint32_t*)calloc(o->slots_cap, sizeof *o->subject_of);
  o->free_at = (uint32_t*)malloc((size_t)o->slots_cap * 4);
  o->G = c.gc_ticks;
  o->base = (uint32_t*)calloc(N, 4); o->base_since = (uint32_t*)calloc(N, 4);
  o->last_change = (uint32_t*)malloc((size_t)N * 4);
  o->C = c.view_cap;
  if (o->C) {
    o->tab = (otab_t*)calloc(N, sizeof *o->tab);
    osent_t* pool = o->tab ? (osent_t*)malloc((size_t)N * o->C * sizeof *pool) : NULL;   /* one block, C entries per member */
    if (!pool) { free(o->tab); o->tab = NULL; }
    else for (uint32_t i = 0; i < N; i++) o->tab[i].v = pool + (size_t)i * o->C;
  }
  pthread_mutex_init(&o->mu, NULL);
  if ((o->C && !o->tab) || !o->up || !o->self_inc || !o->pb || !o->timers || !o->nsent || !o->slot_of || !o->first_suspect || !o->crash_tick ||
      !o->cols || !o->subject_of || !o->free_at || !o->base || !o->base_since || !o->last_change || workers_start(o, 1)) {
    swimoracle_destroy(o); return fail(NULL, SWIMSIM_ERR_NOMEM, "out of memory");
  }
  memset(o->free_at, 0xFF, (size_t)o->slots_cap * 4); memset(o->last_change, 0xFF, (size_t)N * 4);
  memset(o->up, 1, N);
  memset(o->first_suspect, 0xFF, (size_t)N * 4); memset(o->crash_tick, 0xFF, (size_t)N * 4);
  *out = o;
  return SWIMSIM_OK;
}

int swimoracle_create_msg(const swimsim_config_t* cfg, swimoracle_t** out, char* err, size_t errcap) {
  int rc = swimoracle_create(cfg, out);
  if (err && errcap) snprintf(err, errcap, "%s", rc ? g_create_err : "");
  return rc;
}

void swimoracle_destroy(swimoracle_t* o) {
  if (!o) return;
  workers_stop(o);
  for (uint32_t s = 0; s < o->nslots; s++) free(o->cols[s]);
  if (o->timers) for (uint32_t i = 0; i < o->N; i++) free(o->timers[i].v);
  if (o->tab) { free(o->tab[0].v); free(o->tab); }
  free(o->free_at); free(o->base); free(o->base_since); free(o->last_change); free(o->inj);
  free(o->cols); free(o->subject_of); free(o->slot_of); free(o->up); free(o->self_inc); free(o->pb);
  free(o->timers); free(o->nsent); free(o->faults); free(o->first_suspect); free(o->crash_tick);
  free(o->events);
  pthread_mutex_destroy(&o->mu);
  free(o);
}

int swimoracle_inject_rumor(swimoracle_t* o, uint32_t observer, uint32_t subject, uint8_t state, uint32_t incarnation) {
  if (!o) return SWIMSIM_ERR_INVALID;
  if (observer >= o->N || subject >= o->N || state > 2 || incarnation > 0x3FFFFFu) return fail(o, SWIMSIM_ERR_INVALID, "inject_rumor: bad member / state / incarnation");
  if (o->cfg.n_shards > 1) return fail(o, SWIMSIM_ERR_STATE, "inject_rumor: unsharded handles only");
  if (o->C) return fail(o, SWIMSIM_ERR_INVALID, "inject_rumor: not available with bounded member maps (view_cap)");
  if (o->ninj == o->inj_cap) { o->inj_cap = o->inj_cap ? o->inj_cap * 2 : 64; o->inj = (opend_t*)realloc(o->inj, o->inj_cap * sizeof *o->inj); }
  o->inj[o->ninj].dst = observer; o->inj[o->ninj].subject = subject; o->inj[o->ninj].key = key_make(incarnation, state);
  o->ninj++;
  return SWIMSIM_OK;
}

int swimoracle_get_config(const swimoracle_t* o, swimsim_config_t* out) {
  if (!o || !out) return SWIMSIM_ERR_INVALID;
  *out = o->cfg; return SWIMSIM_OK;
}

/* ------------------------------------------------------------------------- */
/* results                                                                    */
/* ------------------------------------------------------------------------- */
static int event_cmp(const void* a, const void* b) {
  const swimsim_event_t* x = (const swimsim_event_t*)a; const swimsim_event_t* y = (const swimsim_event_t*)b;
  if (x->tick != y->tick) return x->tick < y->tick ? -1 : 1;
  if (x->observer != y->observer) return x->observer < y->observer ? -1 : 1;
  if (x->subject != y->subject) return x->subject < y->subject ? -1 : 1;
  uint32_t kx = key_make(x->incarnation, x->state), ky = key_make(y->incarnation, y->state);
  if (kx != ky) return kx < ky ? -1 : 1;
  return 0;
}

int swimoracle_drain_events(swimoracle_t* o, swimsim_event_t* buf, size_t cap, size_t* n_out) {
  if (!o || !n_out) return SWIMSIM_ERR_INVALID;
  qsort(o->events, o->nevents, sizeof *o->events, event_cmp);
  size_t w = 0;
  for (size_t x = 0; x < o->nevents; x++) {            /* collapse (tick,observer,subject) to the final key */
    if (x + 1 < o->nevents && o->events[x + 1].tick == o->events[x].tick &&
        o->events[x + 1].observer == o->events[x].observer && o->events[x + 1].subject == o->events[x].subject) continue;
    o->events[w++] = o->events[x];
  }
  o->nevents = w;
  *n_out = w;
  if (w > cap || (w && !buf)) return SWIMSIM_ERR_BUFFER;
  if (w) memcpy(buf, o->events, w * sizeof *buf);
  o->nevents = 0;
  return SWIMSIM_OK;
}

static int ventry_cmp(const void* a, const void* b) {
  uint32_t x = ((const swimsim_view_entry_t*)a)->subject, y = ((const swimsim_view_entry_t*)b)->subject;
  return x < y ? -1 : x > y;
}

/* `members` (src/Core.hs:76-77): the non-default part of observer's map */
int swimoracle_read_view(swimoracle_t* o, uint32_t observer, swimsim_view_entry_t* buf, size_t cap, size_t* n_out) {
  if (!o || !n_out || observer >= o->N) return SWIMSIM_ERR_INVALID;
  size_t n = 0;
  if (o->C) {                                          /* bounded map: its entries, already sorted by subject */
    const otab_t* tb = &o->tab[observer];
    *n_out = tb->n;
    if (tb->n > cap || (tb->n && !buf)) return SWIMSIM_ERR_BUFFER;
    for (uint32_t x = 0; x < tb->n; x++) {
      memset(&buf[x], 0, sizeof buf[x]);
      buf[x].subject = tb->v[x].subject; buf[x].incarnation = key_inc(tb->v[x].key);
      buf[x].state = (uint8_t)key_state(tb->v[x].key); buf[x].since_tick = tb->v[x].since1 - 1;
    }
    return SWIMSIM_OK;
  }
  for (uint32_t s = 0; s < o->nslots; s++) {
    oentry_t e = o->cols[s][observer];
    if (e.key == 0 || o->subject_of[s] == observer) continue;
    if (n < cap && buf) {
      memset(&buf[n], 0, sizeof buf[n]);
      buf[n].subject = o->subject_of[s]; buf[n].incarnation = key_inc(e.key);
      buf[n].state = (uint8_t)key_state(e.key); buf[n].since_tick = e.since1 - 1;
    }
    n++;
  }
  /* settled subjects: Alive@i (i > 0) stays a listed member, Dead ones were removed (removeDeadNodes) */
  for (uint32_t s = 0; s < o->N; s++) {
    if (!o->base[s] || key_state(o->base[s]) != SWIMSIM_ALIVE || s == observer) continue;
    uint32_t sl = o->slot_of[s];
    if (sl && o->cols[sl - 1][observer].key) continue;      /* listed above with its own entry */
    if (n < cap && buf) {
      memset(&buf[n], 0, sizeof buf[n]);
      buf[n].subject = s; buf[n].incarnation = key_inc(o->base[s]);
      buf[n].state = SWIMSIM_ALIVE; buf[n].since_tick = o->base_since[s];
    }
    n++;
  }
  *n_out = n;
  if (n > cap || (n && !buf)) return SWIMSIM_ERR_BUFFER;
  qsort(buf, n, sizeof *buf, ventry_cmp);
  return SWIMSIM_OK;
}

int swimoracle_read_member(swimoracle_t* o, uint32_t m, swimsim_member_t* out) {
  if (!o || !out || m >= o->N) return SWIMSIM_ERR_INVALID;
  memset(out, 0, sizeof *out);
  out->id = m; out->incarnation = o->self_inc[m]; out->up = o->up[m];
  out->n_rumors = (uint8_t)o->pb[m].n;
  for (int s = 0; s < o->pb[m].n; s++) {
    out->rumors[s].subject = o->pb[m].r[s].subject; out->rumors[s].incarnation = key_inc(o->pb[m].r[s].key);
    out->rumors[s].state = (uint8_t)key_state(o->pb[m].r[s].key); out->rumors[s].tx_left = o->pb[m].r[s].tx;
  }
  /* live timers = entries currently Suspect */
  uint32_t nt = 0;
  if (o->C) for (uint32_t x = 0; x < o->tab[m].n; x++) nt += key_state(o->tab[m].v[x].key) == SWIMSIM_SUSPECT;
  for (uint32_t s = 0; s < o->nslots; s++)
    if (o->subject_of[s] != m && key_state(o->cols[s][m].key) == SWIMSIM_SUSPECT) nt++;
  out->n_timers = (uint16_t)nt;
  return SWIMSIM_OK;
}

int swimoracle_first_detect(swimoracle_t* o, uint64_t* out, size_t n) {
  if (!o || !out || n != o->N) return SWIMSIM_ERR_INVALID;
  for (uint32_t j = 0; j < o->N; j++) out[j] = o->first_suspect[j] == NONE32 ? SWIMSIM_TICK_NONE : o->first_suspect[j];
  return SWIMSIM_OK;
}

/* how far a rumour has got (include/swimsim.h, swimsim_coverage): the up members other than `subject` whose entry about it
 * is at least {incarnation, state} in merge order, and the number of such members at all */
int swimoracle_coverage(swimoracle_t* o, uint32_t subject, uint8_t state, uint32_t incarnation, uint64_t out[2]) {
  if (!o || !out || subject >= o->N || state > SWIMSIM_DEAD || incarnation > INC_MAX) return SWIMSIM_ERR_INVALID;
  const uint32_t key = (incarnation << 2) | state;
  uint64_t hold = 0, up = 0;
  for (uint32_t i = 0; i < o->N; i++) {
    if (i == subject || !o->up[i]) continue;
    up++;
    if (view_get(o, i, subject).key >= key) hold++;
  }
  out[0] = hold; out[1] = up;
  return SWIMSIM_OK;
}

int swimoracle_digest(swimoracle_t* o, uint64_t* out) {
  if (!o || !out) return SWIMSIM_ERR_INVALID;
  uint64_t D = mix64((uint64_t)TAG_TICK + o->tick);
  for (uint32_t i = 0; i < o->N; i++) {
    uint64_t mh = h4(TAG_SELF, i, o->self_inc[i], o->up[i]);
    if (o->C) for (uint32_t x = 0; x < o->tab[i].n; x++) {
      const osent_t* e = &o->tab[i].v[x];
      mh += h4(TAG_VIEW, e->subject, e->key, e->since1);
      if (key_state(e->key) == SWIMSIM_SUSPECT) mh += h4(TAG_TIMER, e->subject, (uint64_t)e->since1 - 1 + o->S, 0);
    }
    for (uint32_t s = 0; s < o->nslots; s++) {
      oentry_t e = o->cols[s][i];
      if (e.key == 0 || o->subject_of[s] == i) continue;
      mh += h4(TAG_VIEW, o->subject_of[s], e.key, e.since1);
      if (key_state(e.key) == SWIMSIM_SUSPECT) mh += h4(TAG_TIMER, o->subject_of[s], (uint64_t)e.since1 - 1 + o->S, 0);
    }
    for (int s = 0; s < o->pb[i].n; s++) mh += h4(TAG_PB, o->pb[i].r[s].subject, o->pb[i].r[s].key, o->pb[i].r[s].tx);
    D += mix64(mh + mix64((uint64_t)TAG_MEMBER + i));
    if (o->first_suspect[i] != NONE32) D += h4(TAG_FD, i, o->first_suspect[i], 0);
    if (o->base[i]) D += h4(TAG_BASE, i, o->base[i], o->base_since[i]);
  }
  *out = D;
  return SWIMSIM_OK;
}

int swimoracle_counters(swimoracle_t* o, uint64_t* out, size_t n) {
  if (!o || !out || n < SWIMSIM_CTR_COUNT) return SWIMSIM_ERR_INVALID;
  memcpy(out, o->counters, sizeof o->counters);
  return SWIMSIM_OK;
}

/* ------------------------------------------------------------------------- */
/* unit-level hooks                                                           */
/* ------------------------------------------------------------------------- */
int swimoracle_k_random_members(swimoracle_t* o, uint32_t observer, uint32_t n, const uint32_t* excludes,
                                size_t n_excludes, uint32_t* out, size_t cap, size_t* n_out) {
  if (!o || !n_out || observer >= o->N || n > 255) return SWIMSIM_ERR_INVALID;
  if (o->C) return fail(o, SWIMSIM_ERR_INVALID, "k_random_members: not available with bounded member maps (view_cap)");
  uint32_t tmp[256];
  o->tk = tick_key(o->cfg.seed, (uint32_t)o->tick);
  uint32_t np = k_random_members(o, observer, n, excludes, n_excludes, P_SELECT, 0, tmp);
  *n_out = np;
  if (np > cap || (np && !out)) return SWIMSIM_ERR_BUFFER;
  memcpy(out, tmp, np * sizeof *out);
  return SWIMSIM_OK;
}

int swimoracle_set_view(swimoracle_t* o, uint32_t observer, uint32_t subject, uint8_t state, uint32_t incarnation) {
  if (!o || observer >= o->N || subject >= o->N || state > 2 || incarnation > INC_MAX || observer == subject)
    return SWIMSIM_ERR_INVALID;
  if (o->C) return fail(o, SWIMSIM_ERR_INVALID, "set_view: not available with bounded member maps (view_cap)");
  oentry_t* e = view_ref(o, observer, subject);
  if (!e) return SWIMSIM_ERR_CAPACITY;
  e->key = key_make(incarnation, state); e->since1 = (uint32_t)o->tick + 1;
  if (state == SWIMSIM_SUSPECT) timer_push(o, observer, subject, (uint32_t)o->tick + o->S);
  return o->poisoned ? SWIMSIM_ERR_CAPACITY : SWIMSIM_OK;
}

int swimoracle_process(swimoracle_t* o, uint32_t self, uint32_t sender, const swimoracle_msg_t* msg,
                       int literal_d8, swimoracle_msg_t* out, size_t cap, size_t* n_out) {
  if (!o || !msg || !n_out || self >= o->N) return SWIMSIM_ERR_INVALID;
  omsg_t m; memset(&m, 0, sizeof m);
  m.m = *msg; m.relay_to = NONE32;
  o->capture = 1; o->capture_literal_d8 = literal_d8; o->cap_out = out; o->cap_cap = out ? cap : 0; o->cap_n = 0;
  process(&o->ctx[0], self, sender, &m);
  fold_workers(o);
  o->capture = 0;
  *n_out = o->cap_n;
  return o->cap_n > o->cap_cap ? SWIMSIM_ERR_BUFFER : SWIMSIM_OK;
}

/* literal suspectOrDeadNode' on one entry (src/Core.hs:142-187), subject != self, known */
uint32_t swimoracle_reference_rule(uint32_t cur_key, uint32_t msg_key) {
  uint32_t i = key_inc(msg_key), st = key_state(msg_key);
  uint32_t inc = key_inc(cur_key), cs = key_state(cur_key);
  if (st == SWIMSIM_ALIVE) return cur_key;                    /* aliveNode is unwritten (D6) */
  if (i < inc) return cur_key;                                /* :151 `i < memberIncarnation m` */
  if (st == SWIMSIM_SUSPECT && cs != SWIMSIM_ALIVE) return cur_key;   /* livenessCheck IsSuspect :183 */
  if (st == SWIMSIM_DEAD && cs == SWIMSIM_DEAD) return cur_key;       /* livenessCheck IsDead :184 */
  return key_make(i, st);                                     /* :169-176 */
}

uint32_t swimoracle_merge_rule(uint32_t cur_key, uint32_t msg_key) {
  return msg_key > cur_key ? msg_key : cur_key;
}

/* removeDeadNodes (src/Core.hs:65-67): Map.filter (not . isDead) */
size_t swimoracle_remove_dead_nodes(swimsim_view_entry_t* entries, size_t n) {
  size_t w = 0;
  for (size_t x = 0; x < n; x++) if (entries[x].state != SWIMSIM_DEAD) entries[w++] = entries[x];
  return w;
}

/* oracle-only: step under the LITERAL suspectOrDeadNode' (src/Core.hs:151-152,182-184) instead of the
 * commutative merge; *hits = proposals on which the two rules disagreed so far (D13).  While it stays 0 a
 * run is identical to the merge run -- which is how the tests narrow "parity unpinned". */
int swimoracle_set_literal_rule(swimoracle_t* o, int on) {
  if (!o || (on && (o->C || o->cfg.gc_ticks || o->cfg.join_pull || o->cfg.pull_ticks))) return SWIMSIM_ERR_INVALID;
  o->literal_rule = on != 0; return SWIMSIM_OK;
}
uint64_t swimoracle_d13_hits(const swimoracle_t* o) { return o ? o->d13_hits : 0; }

int swimoracle_set_shuffle(swimoracle_t* o, uint64_t shuffle_seed) {
  if (!o) return SWIMSIM_ERR_INVALID;
  o->shuffle_seed = shuffle_seed; return SWIMSIM_OK;
}
