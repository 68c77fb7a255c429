// swim_kernels.h -- the per-tick HIP kernels (gfx950).  Integer / indexing work; no MFMA on
// this path.  The binding resource is the number of scattered L2 requests per member-tick
// (measured: profiles/), so the layout keeps ONE gathered word per probe target (minfo), the
// receiver filters incoming rumours against its own piggyback buffer in registers before it
// touches its view row, and only cross-member deliveries go through atomics.
// Two launches per tick:
//   probe_kernel : one period of failureDetector / probeNode' per member (src/Core.hs:233-269),
//                  closed form of the message exchange; emits (dst <- src) payload deliveries.
//   merge_kernel : owner-computes end of tick: timers, state rule, piggyback queue
//                  (src/Core.hs:89-117, 127-138, 142-218).
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

__device__ inline unsigned wave_sum(unsigned x) {
  for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, 64);
  return x;
}

__device__ inline void ctr_flush(const DevState& s, BlockCounters* sh, uint32_t row) {
  __syncthreads();
  if (threadIdx.x < C_COUNT) {
    unsigned long long x = threadIdx.x == C_EVDIGEST ? sh->evd : (unsigned long long)sh->v[threadIdx.x];
    if (x) s.blk[(size_t)row * C_COUNT + threadIdx.x] += x;
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

// make sure subject j has a rumour slot (first Suspect about j); returns nothing: the slot is
// only needed by the NEXT kernel
__device__ inline void ensure_slot(const DevState& s, uint32_t j) {
  uint32_t cur = s.minfo[j];
  while ((cur & MI_SLOT) == 0u) {
    const uint32_t seen = atomicCAS(&s.minfo[j], cur, cur | MI_SLOT);   // 0xFFFF = being allocated
    if (seen == cur) {
      uint32_t r = atomicAdd(&s.g[G_NSLOTS], 1u);
      if (r >= s.R_max) { atomicOr(&s.g[G_ERR], (uint32_t)ERRF_SUBJECTS); r = 0; }
      s.subject_of[r] = j;
      __threadfence();
      atomicXor(&s.minfo[j], MI_SLOT ^ (r + 1u));
      return;
    }
    cur = seen;
  }
}

// ================================================================================================
// probe kernel
// ================================================================================================
template <int PMAX>
__global__ __launch_bounds__(BLOCK) void probe_kernel(DevState s, uint32_t t, uint32_t tk) {
  __shared__ BlockCounters sh;
  ctr_init(&sh);
  const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
  const uint32_t mi = i < s.N ? s.minfo[i] : 0u;
  const bool act = mi_up(mi);
  unsigned n_pings = 0;
  if (act) {
    const uint32_t mk = mix32(tk ^ i);
    const uint32_t mycnt = mi_pbn(mi);
    const uint32_t mysrc = mi_src(i, mi);
    uint32_t picks[PMAX], pinfo[PMAX];
    // ms <- kRandomMembers store (numToGossip cfg) []        (src/Core.hs:239)
    const uint32_t np = select_members<PMAX>(s, mk, i, s.P, P_SELECT, 0, nullptr, 0, picks, pinfo);
    n_pings = np;
    uint32_t nfail = 0, nack = 0;
    unsigned payloads = 0, rumors = 0, dfail = 0, preqs = 0, susp = 0, fsusp = 0;
    // pass 1: outcome of every direct probe -- pure arithmetic on the gathered info words.
    //   Direct (Ping seq j) is delivered iff not lost and j is up (src/Core.hs:246);
    //   j answers Ack (src/Core.hs:97-99), which may be lost too.
    bool ping_ok[PMAX], ack_ok[PMAX];
#pragma unroll
    for (int p = 0; p < PMAX; ++p) {
      ping_ok[p] = false; ack_ok[p] = false;
      if ((uint32_t)p < np) {
        ping_ok[p] = mi_up(pinfo[p]) && !lost(s, tk, P_L_PING, i, picks[p], p);
        ack_ok[p] = ping_ok[p] && !lost(s, tk, P_L_ACK, picks[p], i, p);
      }
    }
    // pass 2: the Pings' piggyback payloads: one delivery record per target.  The reservation
    // atomics are independent, so they are issued back to back and overlap in the fabric.
    if (mycnt) {
      uint32_t pos[PMAX];
#pragma unroll
      for (int p = 0; p < PMAX; ++p) { pos[p] = 0; if (ping_ok[p]) pos[p] = atomicAdd(&s.inbox_cnt[picks[p]], 1u); }
#pragma unroll
      for (int p = 0; p < PMAX; ++p)
        if (ping_ok[p]) { push_commit(s, t, picks[p], mysrc, pos[p]); payloads++; rumors += mycnt; }
    }
    // pass 3: the Acks' payloads are pulled by the prober itself: private list, no atomics
#pragma unroll
    for (int p = 0; p < PMAX; ++p) {
      const uint32_t pj = mi_pbn(pinfo[p]);
      if (ack_ok[p] && pj) {
        s.ackfrom[(size_t)i * s.P + nack] = mi_src(picks[p], pinfo[p]);
        nack++; payloads++; rumors += pj;
      }
    }
    // pass 4 (rare): probes without an ack -> k indirect probes -> maybe Suspect
    for (int p = 0; p < PMAX; ++p) {
      if ((uint32_t)p >= np) break;
      if (ack_ok[p]) continue;                               // unlessAck (D2, D3)
      const uint32_t j = picks[p], mj = pinfo[p];
      const bool upj = mi_up(mj);
      const uint32_t pj = mi_pbn(mj);
      dfail++;
      // kRandomMembers store (numToGossip cfg) [] for proxies (src/Core.hs:249), D7: not the target
      uint32_t qs[PMAX], qinfo[PMAX];
      const uint32_t excl = j;
      const uint32_t nq = select_members<PMAX>(s, mk, i, s.K, P_PROXY, p, &excl, 1, qs, qinfo);
      preqs += nq;
      bool acked = false;
      for (int k = 0; k < PMAX; ++k) {
        if ((uint32_t)k >= nq) break;
        const uint32_t q = qs[k], mq = qinfo[k];
        const uint32_t pq = mi_pbn(mq);
        const uint32_t idx = ((uint32_t)p << 8) | (uint32_t)k;
        // i -> q : IndirectPing (src/Core.hs:250, 262-269)
        if (lost(s, tk, P_L_REQ, i, q, idx) || !mi_up(mq)) continue;
        if (mycnt) { push(s, t, q, mysrc); payloads++; rumors += mycnt; }
        // q -> j : Ping on behalf of i (src/Core.hs:105-108; D8, D12)
        if (!upj || lost(s, tk, P_L_FWD, q, j, idx)) continue;
        if (pq) { push(s, t, j, mi_src(q, mq)); payloads++; rumors += pq; }
        // j -> q : Ack
        if (lost(s, tk, P_L_BACK, j, q, idx)) continue;
        if (pj) { push(s, t, q, mi_src(j, mj)); payloads++; rumors += pj; }
        // q -> i : relayed Ack (D9)
        if (lost(s, tk, P_L_RELAY, q, i, idx)) continue;
        if (pq) { push(s, t, i, mi_src(q, mq)); payloads++; rumors += pq; }
        acked = true;
      }
      if (acked) continue;                                   // second unlessAck (src/Core.hs:251)
      // suspectNode store (Suspect (memberIncarnation m) name)  (src/Core.hs:253): lands in merge
      ensure_slot(s, j);
      s.fail[(size_t)i * s.P + nfail] = j;
      nfail++;
      susp++;
      if (upj) fsusp++;
      else atomicMin(&s.first_suspect[j], t);
    }
    s.probe_out[i] = (uint16_t)(np | (nfail << 5) | (nack << 10));
    ctr_add(&sh, C_PAYLOADS, payloads);
    ctr_add(&sh, C_RUMORS_SEEN, rumors);
    ctr_add(&sh, C_DIRECT_FAILED, dfail);
    ctr_add(&sh, C_PING_REQS, preqs);
    ctr_add(&sh, C_SUSPECTS, susp);
    ctr_add(&sh, C_FALSE_SUSPECTS, fsusp);
  }
  // the two always-nonzero counters: wave-reduce first
  unsigned wp = wave_sum(n_pings);
  unsigned wa = wave_sum(act ? 1u : 0u);
  if ((threadIdx.x & 63) == 0) { ctr_add(&sh, C_PINGS, wp); ctr_add(&sh, C_ACTIVE, wa); }
  ctr_flush(s, &sh, blockIdx.x);
}

// ================================================================================================
// merge kernel
// ================================================================================================

// candidate set for the new piggyback buffer: PB_SLOTS best by (tx desc, subject asc) (H3, D5)
struct Cand {
  uint32_t slot[PB_SLOTS], key[PB_SLOTS], tx[PB_SLOTS], subj[PB_SLOTS];
  uint32_t n;
};

__device__ inline bool rumor_better(uint32_t txa, uint32_t sa, uint32_t txb, uint32_t sb) {
  return txa != txb ? txa > txb : sa < sb;
}

__device__ inline void cand_insert(Cand& c, uint32_t slot, uint32_t subj, uint32_t key, uint32_t tx) {
  bool done = false;
#pragma unroll
  for (int k = 0; k < PB_SLOTS; ++k)
    if (!done && (uint32_t)k < c.n && c.slot[k] == slot) { c.key[k] = key; c.tx[k] = tx; done = true; }
  if (done) return;
  if (c.n < (uint32_t)PB_SLOTS) {
#pragma unroll
    for (int k = 0; k < PB_SLOTS; ++k)
      if ((uint32_t)k == c.n) { c.slot[k] = slot; c.key[k] = key; c.tx[k] = tx; c.subj[k] = subj; }
    c.n++;
    return;
  }
  uint32_t wtx = c.tx[0], ws = c.subj[0]; int worst = 0;
#pragma unroll
  for (int k = 1; k < PB_SLOTS; ++k)
    if (rumor_better(wtx, ws, c.tx[k], c.subj[k])) { worst = k; wtx = c.tx[k]; ws = c.subj[k]; }
  if (rumor_better(tx, subj, wtx, ws)) {
#pragma unroll
    for (int k = 0; k < PB_SLOTS; ++k)
      if (k == worst) { c.slot[k] = slot; c.key[k] = key; c.tx[k] = tx; c.subj[k] = subj; }
  }
}

constexpr int SEEN = 4;      // rumours looked up this tick (register cache of (slot, view key))
constexpr int ACC_CAP = 8;   // per-thread list of accepted changes awaiting the apply phase (LDS)

struct MergeCtx {
  uint32_t i, t;
  uint4 hot;
  Cand c;
  uint32_t seen_slot[SEEN], seen_key[SEEN], seen_pos;
  uint32_t nacc;
  unsigned changes, timers_fired, evdropped;
  unsigned long long evd, ha;
};

__device__ inline void emit_event(const DevState& s, MergeCtx& m, uint32_t observer, uint32_t subject,
                                  uint32_t key, uint32_t cause) {
  if (!(s.event_mask & (1u << cause))) return;
  uint32_t pos = atomicAdd(&s.g[G_EVCUR], 1u);
  if (pos < s.event_cap) s.events[pos] = make_uint4(m.t, observer, subject, (key << 8) | cause);
  else m.evdropped++;
}

// Apply phase for one accepted change (heavy, runs convergently over the short per-thread list):
// bookkeeping that follows `saveMember m'` in suspectOrDeadNode' (src/Core.hs:169-179): timer
// start (D4), enqueue for piggybacking (D5), membership event, digest.
__device__ inline void apply_change(const DevState& s, MergeCtx& m, uint32_t w0, uint32_t key, uint32_t oldkey) {
  const uint32_t slot = w0 & 0xFFFFu, cause = (w0 >> 16) & 3u, first = (w0 >> 18) & 1u;
  const uint32_t subject = s.subject_of[slot];
  const unsigned long long x = mix64(m.ha + subject);          // h4(TAG_EV, a, subject, .) prefix
  m.evd += mix64(x + key) - mix64(x + oldkey);
  m.changes += first;
  if (cause == 1u) m.timers_fired++;
  if ((key & 3u) == ST_SUSPECT) {                              // start the suspicion timer (D4)
    if (m.hot.z >= s.timer_cap) atomicOr(&s.g[G_ERR], (uint32_t)ERRF_TIMERS);
    else {
      uint32_t pos = m.hot.y + m.hot.z; if (pos >= s.timer_cap) pos -= s.timer_cap;
      s.ring[(size_t)m.i * s.timer_cap + pos] = make_uint2(slot, m.t + s.S);
      if (m.hot.z == 0) m.hot.w = m.t + s.S;
      m.hot.z++;
    }
  }
  cand_insert(m.c, slot, subject, key, s.L);                   // `Just msg` -> Broadcast -> enqueue (D5)
  emit_event(s, m, m.i, subject, key, cause);
}

__global__ __launch_bounds__(BLOCK) void merge_kernel(DevState s, uint32_t t) {
  __shared__ BlockCounters sh;
  __shared__ uint32_t acc[ACC_CAP][3][BLOCK];                  // [entry][word][thread]: conflict-free
  ctr_init(&sh);
  const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
  const uint32_t tid = threadIdx.x;
  if (blockIdx.x == 0 && threadIdx.x == 0) s.g[G_OVF0 + ((t + 1) & 1u)] = 0;  // next tick's overflow list
  const uint32_t mi = i < s.N ? s.minfo[i] : 0u;
  if (mi_up(mi)) {
    const uint32_t po = s.probe_out[i];
    const uint32_t nsent = po & 31u, nfail = (po >> 5) & 31u, nack = po >> 10;
    const uint32_t cnt = s.inbox_cnt[i];
    MergeCtx m;
    m.i = i; m.t = t; m.hot = s.hot[i];
    const uint32_t pcount = mi_pbn(mi), cur = mi_buf(mi);
    const bool timer_due = m.hot.z && m.hot.w <= t;
    if (cnt | nfail | nack | pcount | (uint32_t)timer_due) {
      m.c.n = 0; m.changes = 0; m.timers_fired = 0; m.evdropped = 0; m.evd = 0; m.seen_pos = 0; m.nacc = 0;
      m.ha = 0;
#pragma unroll
      for (int k = 0; k < SEEN; ++k) { m.seen_slot[k] = NONE32; m.seen_key[k] = 0; }
      const uint4 hot0 = m.hot;
      // age the queue: every ping sent this tick carried every slot (D5)
      if (pcount) {
        const uint4* line = reinterpret_cast<const uint4*>(s.pb + ((size_t)cur * s.N + i) * PB_SLOTS);
        uint4 v[PB_SLOTS / 2];
#pragma unroll
        for (int h = 0; h < PB_SLOTS / 2; ++h) v[h] = line[h];
#pragma unroll
        for (int h = 0; h < PB_SLOTS / 2; ++h) {
          const uint32_t tx0 = (v[h].x >> 16) & 0xFFu, tx1 = (v[h].z >> 16) & 0xFFu;
          if (tx0 > nsent) cand_insert(m.c, v[h].x & 0xFFFFu, s.subject_of[v[h].x & 0xFFFFu], v[h].y, tx0 - nsent);
          if (tx1 > nsent) cand_insert(m.c, v[h].z & 0xFFFFu, s.subject_of[v[h].z & 0xFFFFu], v[h].w, tx1 - nsent);
        }
      }
      const uint32_t my_slot1 = mi & MI_SLOT;                   // slot+1 of rumours about me
      const uint32_t self_inc0 = m.hot.x;
      uint32_t refute = NONE32;

      auto apply_all = [&]() {
        if (m.nacc && !m.ha) m.ha = mix64(mix64((uint64_t)TAG_EV) + (((uint64_t)t << 32) | i));
        for (uint32_t k = 0; k < m.nacc; ++k) apply_change(s, m, acc[k][0][tid], acc[k][1][tid], acc[k][2][tid]);
        m.nacc = 0;
      };
      // The state rule: suspectOrDeadNode' (src/Core.hs:142-187) + the unwritten aliveNode
      // (:197-218, D6) as the commutative merge entry := max(entry, (incarnation,state)) (H3,
      // D13).  Scan phase: the entry is updated at once (memberLastChange = now, :176) and the
      // change is queued; everything else happens in apply_change.  Caller checked key > e.x.
      auto note = [&](uint32_t slot, uint32_t key, uint32_t cause, uint2 e) {
        s.V[vidx(s, i, slot)] = make_uint2(key, t + 1);
        if (m.nacc == (uint32_t)ACC_CAP) apply_all();
        acc[m.nacc][0][tid] = slot | (cause << 16) | ((e.y != t + 1 ? 1u : 0u) << 18);
        acc[m.nacc][1][tid] = key;
        acc[m.nacc][2][tid] = e.x;
        m.nacc++;
#pragma unroll
        for (int k = 0; k < SEEN; ++k) if ((uint32_t)k == m.seen_pos) { m.seen_slot[k] = slot; m.seen_key[k] = key; }
        m.seen_pos = (m.seen_pos + 1) & (SEEN - 1);
      };

      // phase 1: suspicion timers (the FIXME at src/Core.hs:141; D4)
      while (m.hot.z) {
        const uint2 tm = s.ring[(size_t)i * s.timer_cap + m.hot.y];
        if (tm.y > t) break;
        m.hot.y = (m.hot.y + 1 == s.timer_cap) ? 0 : m.hot.y + 1;
        m.hot.z--;
        const uint2 e = s.V[vidx(s, i, tm.x)];
        if ((e.x & 3u) == ST_SUSPECT && e.y - 1 + s.S == tm.y) note(tm.x, (e.x & ~3u) | ST_DEAD, 1u /*TIMER*/, e);
      }
      // phase 2: own probes that ended without any ack: Suspect at the viewed incarnation
      for (uint32_t f = 0; f < nfail; ++f) {
        const uint32_t j = s.fail[(size_t)i * s.P + f];
        const uint32_t sl = (s.minfo[j] & MI_SLOT) - 1;
        const uint2 e = s.V[vidx(s, i, sl)];
        const uint32_t key = (e.x & ~3u) | ST_SUSPECT;
        if (key > e.x) note(sl, key, 0u /*PROBE*/, e);
      }
      // phase 3: rumours received this tick (any order: the merge is commutative)
      auto take = [&](uint32_t srcw) {
        const uint32_t src = srcw & 0x7FFFFFFFu, buf = srcw >> 31;
        const uint4* line = reinterpret_cast<const uint4*>(s.pb + ((size_t)buf * s.N + src) * PB_SLOTS);
        uint4 v[PB_SLOTS / 2];
#pragma unroll
        for (int h = 0; h < PB_SLOTS / 2; ++h) v[h] = line[h];
#pragma unroll
        for (int h = 0; h < PB_SLOTS / 2; ++h) {
#pragma unroll
          for (int w = 0; w < 2; ++w) {
            const uint32_t lo = w ? v[h].z : v[h].x, key = w ? v[h].w : v[h].y;
            if (!((lo >> 16) & 0xFFu)) continue;
            const uint32_t sl = lo & 0xFFFFu;
            if (sl + 1 == my_slot1) {
              // about self -> refute (src/Core.hs:155-166); old incarnations ignored (:151)
              if ((key & 3u) != ST_ALIVE && (key >> 2) >= self_inc0)
                refute = (refute == NONE32 || (key >> 2) > refute) ? (key >> 2) : refute;
              continue;
            }
            // already known?  my own queue and this tick's lookups are in registers: no memory touch
            bool known = false;
#pragma unroll
            for (int k = 0; k < PB_SLOTS; ++k) known |= ((uint32_t)k < m.c.n) && m.c.slot[k] == sl && m.c.key[k] >= key;
#pragma unroll
            for (int k = 0; k < SEEN; ++k) known |= m.seen_slot[k] == sl && m.seen_key[k] >= key;
            if (known) continue;
            const uint2 e = s.V[vidx(s, i, sl)];
            if (key > e.x) note(sl, key, 2u /*GOSSIP*/, e);
            else {
#pragma unroll
              for (int k = 0; k < SEEN; ++k) if ((uint32_t)k == m.seen_pos) { m.seen_slot[k] = sl; m.seen_key[k] = e.x; }
              m.seen_pos = (m.seen_pos + 1) & (SEEN - 1);
            }
          }
        }
      };
      for (uint32_t x = 0; x < nack; ++x) take(s.ackfrom[(size_t)i * s.P + x]);
      const uint32_t nin = cnt < s.inbox_cap ? cnt : s.inbox_cap;
      for (uint32_t x = 0; x < nin; ++x) take(s.inbox[(size_t)i * s.inbox_cap + x]);
      if (cnt > s.inbox_cap) {
        const uint32_t no = min(s.g[G_OVF0 + (t & 1u)], s.ovf_cap);
        for (uint32_t x = 0; x < no; ++x) {
          const uint2 o = s.ovf[(size_t)(t & 1u) * s.ovf_cap + x];
          if (o.x == i) take(o.y);
        }
      }
      apply_all();
      // refutation: bump own incarnation past the rumour's (src/Core.hs:155-166; D10)
      unsigned refutes = 0;
      if (refute != NONE32) {
        uint32_t ni = refute + 1;
        if (ni > INC_MAX) { atomicOr(&s.g[G_ERR], (uint32_t)ERRF_INC); ni = INC_MAX; }
        m.hot.x = ni;
        m.evd += h4(TAG_INC, ((uint64_t)t << 32) | i, ni, 0);
        refutes = 1;
        cand_insert(m.c, my_slot1 - 1, i, (ni << 2) | ST_ALIVE, s.L);   // Just Alive{..} (:163)
        emit_event(s, m, i, i, (ni << 2) | ST_ALIVE, 3u /*REFUTE*/);
      }
      // write back
      if (m.c.n) {
        uint4* line = reinterpret_cast<uint4*>(s.pb + ((size_t)(cur ^ 1u) * s.N + i) * PB_SLOTS);
#pragma unroll
        for (int h = 0; h < PB_SLOTS / 2; ++h) {
          uint4 v;
          v.x = (uint32_t)(2 * h) < m.c.n ? (m.c.slot[2 * h] | (m.c.tx[2 * h] << 16)) : 0u;
          v.y = (uint32_t)(2 * h) < m.c.n ? m.c.key[2 * h] : 0u;
          v.z = (uint32_t)(2 * h + 1) < m.c.n ? (m.c.slot[2 * h + 1] | (m.c.tx[2 * h + 1] << 16)) : 0u;
          v.w = (uint32_t)(2 * h + 1) < m.c.n ? m.c.key[2 * h + 1] : 0u;
          line[h] = v;
        }
        s.minfo[i] = (mi & ~MI_PB) | (m.c.n << MI_PBN_SHIFT) | ((cur ^ 1u) << 20);
      } else if (pcount) {
        s.minfo[i] = mi & ~MI_PB;
      }
      if (m.hot.z == 0) m.hot.w = NONE32;
      else if (m.hot.y != hot0.y) m.hot.w = s.ring[(size_t)i * s.timer_cap + m.hot.y].y;
      if (m.hot.x != hot0.x || m.hot.y != hot0.y || m.hot.z != hot0.z || m.hot.w != hot0.w) s.hot[i] = m.hot;
      if (cnt) s.inbox_cnt[i] = 0;
      ctr_add(&sh, C_CHANGES, m.changes);
      ctr_add(&sh, C_PB_WRITES, (pcount || m.c.n) ? 1u : 0u);
      ctr_add(&sh, C_TIMERS_FIRED, m.timers_fired);
      ctr_add(&sh, C_REFUTES, refutes);
      ctr_add(&sh, C_EVENTS_DROPPED, m.evdropped);
      if (m.evd) atomicAdd(&sh.evd, m.evd);
    }
  }
  ctr_flush(s, &sh, blockIdx.x);
}

// ================================================================================================
// auxiliary kernels
// ================================================================================================
struct FaultRec { uint32_t member, up; };

// ground-truth changes for tick t, applied in order by one thread (few per tick)
__global__ void fault_kernel(DevState s, uint32_t t, const FaultRec* faults, uint32_t nfaults) {
  if (blockIdx.x || threadIdx.x) return;
  unsigned long long evd = 0; unsigned dropped = 0;
  for (uint32_t k = 0; k < nfaults; ++k) {
    const uint32_t mbr = faults[k].member, up = faults[k].up;
    uint32_t mi = s.minfo[mbr];
    if ((uint32_t)mi_up(mi) == up) continue;
    s.first_suspect[mbr] = NONE32;
    if (!up) { s.minfo[mbr] = mi & ~MI_UP; s.crash_tick[mbr] = t; continue; }
    // (re)join: new incarnation, announce Alive
    uint4 hot = s.hot[mbr];
    uint32_t ni = hot.x + 1;
    if (ni > INC_MAX) { atomicOr(&s.g[G_ERR], (uint32_t)ERRF_INC); ni = INC_MAX; }
    hot.x = ni; s.hot[mbr] = hot;
    evd += h4(TAG_INC, ((uint64_t)t << 32) | mbr, ni, 0);
    ensure_slot(s, mbr);
    mi = s.minfo[mbr];
    const uint32_t sl = (mi & MI_SLOT) - 1;
    const uint32_t pcount = mi_pbn(mi), cur = mi_buf(mi);
    Cand c; c.n = 0;
    uint64_t* line = s.pb + ((size_t)cur * s.N + mbr) * PB_SLOTS;
    if (pcount)
      for (int q = 0; q < PB_SLOTS; ++q) {
        const uint32_t lo = (uint32_t)line[q], hi = (uint32_t)(line[q] >> 32);
        if ((lo >> 16) & 0xFFu) cand_insert(c, lo & 0xFFFFu, s.subject_of[lo & 0xFFFFu], hi, (lo >> 16) & 0xFFu);
      }
    cand_insert(c, sl, mbr, (ni << 2) | ST_ALIVE, s.L);
    for (int q = 0; q < PB_SLOTS; ++q) {
      uint64_t v = 0;
      for (int k2 = 0; k2 < PB_SLOTS; ++k2)
        if (k2 == q && (uint32_t)q < c.n) v = ((uint64_t)c.key[k2] << 32) | (c.slot[k2] | (c.tx[k2] << 16));
      line[q] = v;
    }
    s.minfo[mbr] = (mi & ~MI_PBN) | (c.n << MI_PBN_SHIFT) | MI_UP;
    if (s.event_mask & (1u << 4)) {
      uint32_t pos = atomicAdd(&s.g[G_EVCUR], 1u);
      if (pos < s.event_cap) s.events[pos] = make_uint4(t, mbr, mbr, (((ni << 2) | ST_ALIVE) << 8) | 4u);
      else dropped++;
    }
  }
  s.blk[(size_t)s.nblocks * C_COUNT + C_EVDIGEST] += evd;
  s.blk[(size_t)s.nblocks * C_COUNT + C_EVENTS_DROPPED] += dropped;
}

// full-state digest: Sum_i mix64(member_hash(i) + mix64(TAG_MEMBER + i)) + first-detection terms
__global__ __launch_bounds__(BLOCK) void digest_kernel(DevState s, unsigned long long* out) {
  __shared__ unsigned long long acc;
  if (threadIdx.x == 0) acc = 0;
  __syncthreads();
  const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
  if (i < s.N) {
    const uint4 hot = s.hot[i];
    const uint32_t mi = s.minfo[i];
    unsigned long long mh = h4(TAG_SELF, i, hot.x, mi_up(mi) ? 1u : 0u);
    const uint32_t ns = min(s.g[G_NSLOTS], s.R_max);
    for (uint32_t r = 0; r < ns; ++r) {
      const uint2 e = s.V[vidx(s, i, r)];
      if (e.x == 0) continue;
      const uint32_t subject = s.subject_of[r];
      if (subject == i) continue;
      mh += h4(TAG_VIEW, subject, e.x, e.y);
      if ((e.x & 3u) == ST_SUSPECT) mh += h4(TAG_TIMER, subject, (uint64_t)e.y - 1 + s.S, 0);
    }
    if (mi_pbn(mi)) {
      const uint64_t* line = s.pb + ((size_t)mi_buf(mi) * s.N + i) * PB_SLOTS;
      for (int q = 0; q < PB_SLOTS; ++q) {
        const uint32_t lo = (uint32_t)line[q], hi = (uint32_t)(line[q] >> 32);
        if ((lo >> 16) & 0xFFu) mh += h4(TAG_PB, s.subject_of[lo & 0xFFFFu], hi, (lo >> 16) & 0xFFu);
      }
    }
    unsigned long long d = mix64(mh + mix64((uint64_t)TAG_MEMBER + i));
    const uint32_t fs = s.first_suspect[i];
    if (fs != NONE32) d += h4(TAG_FD, i, fs, 0);
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
  s.V[vidx(s, observer, sl)] = make_uint2(key, t + 1);
  if ((key & 3u) == ST_SUSPECT) {
    uint4 hot = s.hot[observer];
    if (hot.z >= s.timer_cap) { atomicOr(&s.g[G_ERR], (uint32_t)ERRF_TIMERS); return; }
    uint32_t pos = hot.y + hot.z; if (pos >= s.timer_cap) pos -= s.timer_cap;
    s.ring[(size_t)observer * s.timer_cap + pos] = make_uint2(sl, t + s.S);
    if (hot.z == 0) hot.w = t + s.S;
    hot.z++;
    s.hot[observer] = hot;
  }
}

__global__ void init_members_kernel(uint4* hot, uint32_t* minfo, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { hot[i] = make_uint4(0u, 0u, 0u, NONE32); minfo[i] = MI_UP; }
}

}  // namespace swim
