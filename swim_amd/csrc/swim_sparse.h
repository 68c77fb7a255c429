// swim_sparse.h -- the tick with BOUNDED member maps (view_cap = C > 0; include/swimsim.h "Bounded member maps",
// DESIGN.md section 2.8): what runs BASELINE config 5 (30 % message loss x churn) at millions of members per GPU, where
// nearly every member is the subject of somebody's false suspicion all the time and a dense view row per subject
// (swim_kernels.h) cannot exist.
//
// A member's map is `Map String Member` (src/Types.hs:55) with a capacity: <= C exceptions {subject, key, lastChange} to the
// default "Alive at incarnation 0", C x 12 bytes per member in HBM.
//   * sp_probe_lane_kernel : one member per LANE; "is c Alive in my view" (kRandomMembers, src/Core.hs:72-74) is one bit of the
//     member's filter (written with the map) and, for the few set bits, a look into the map by the whole wave; every delivery an
//     inbox append (an atomic with a returned position).  (sp_probe_kernel, the first form -- one WAVE per member, the map in the
//     wave's registers -- stays for A/B: SWIMSIM_SP_PROBE=wave; it was bound by its wave-uniform instructions, 3.2 against 1.2 ms);
//   * sp_merge_kernel : ONE WAVE steps one member's end of tick -- the 64-wide wavefront is the unit of work there, not the
//     thread: the member's map is rebuilt in LDS as a hash table; the lanes are the RUMOURS of the tick (8 sources x
//     8 queue entries per round of loads) and the state rule suspectOrDeadNode' / aliveNode (src/Core.hs:142-218) is an LDS
//     atomicMax on the packed (incarnation, state) key -- the commutative merge of H3 / D13 in one instruction; who stays
//     when the map is over capacity is a radix select over (lastChange, rank) in LDS; the map streams back coalesced.
// Per member-tick at 30 % loss: ~190 rumours in, ~75 subjects the member did not know, 3 KB of map read and written.
// The rules themselves are the unbounded tick's (swim_kernels.h cites them per line); what differs is stated over SETS in
// include/swimsim.h, because a capacity makes "one proposal after the other" depend on the order.
#pragma once
#include "swim_kernels.h"

namespace swim {

constexpr uint32_t SP_WAVES = BLOCK / 64;      // members a workgroup steps at a time
constexpr uint32_t SP_PRIO_TIMER = 2u, SP_PRIO_PROBE = 1u, SP_PRIO_GOSSIP = 0u;   // who states a key first (phase order)
constexpr uint32_t SP_KEPT = 1u << 30;         // h0 bit (after the selection): the entry stays in the map
// A member's "who is NOT Alive in my view" filter: 16 bits per map entry of the capacity (1 024 bits for C <= 64, 2 048 for C <= 128,
// 4 096 beyond: s.sp_bloom_log2 = 10 / 11 / 12), one hash; rewritten by sp_merge_kernel with the map, read by sp_probe_lane_kernel (a
// set bit = look the subject up in the map; a clear one = Alive for sure: no entry, or an Alive one)
constexpr uint32_t SP_BLOOM_WORDS_MAX = 128u;
__device__ inline uint32_t sp_bloom_words(const DevState& s) { return 1u << (s.sp_bloom_log2 - 5u); }
__device__ inline uint32_t sp_bloom_bit(const DevState& s, uint32_t subject) { return (subject * 0x9E3779B1u) >> (32u - s.sp_bloom_log2); }

// ---- layout helpers ----------------------------------------------------------------------------------
// tab[N][3][C]: subjects, keys, lastChange + 1 of member li, entry e -- three coalesced runs per member
__device__ inline uint32_t* sp_row(const DevState& s, uint32_t li, uint32_t field) { return s.sp_tab + ((size_t)li * 3u + field) * s.C; }
// queue lines sq[2][N][8] {subject, key | tx << 24}: buffer (t & 1) is read in tick t, the other one written
__device__ inline uint2* sp_line(const DevState& s, uint32_t buf, uint32_t li) { return s.sp_q + ((size_t)buf * s.N + li) * PB_SLOTS; }
// the queue line of member g (GLOBAL id) as the sources of tick t see it: on one handle the current buffer itself; on a shard the
// replica sp_qall[NT][8] of everybody's start-of-tick line (own slice copied in by sp_publish_kernel, the others all-gathered)
__device__ inline const uint2* sp_src_line(const DevState& s, uint32_t cur, uint32_t g) {
  return s.n_shards > 1 ? s.sp_qall + (size_t)g * PB_SLOTS : s.sp_q + ((size_t)cur * s.N + g) * PB_SLOTS;
}
// sb[NT]: bit 0 up (ground truth), bits 1-4 queue length -- the one byte a prober gathers about a target (the `mb` table)
__device__ inline uint32_t sb_up(uint32_t b) { return b & MB_UP; }
__device__ inline uint32_t sb_qn(uint32_t b) { return (b >> MB_PBN_SHIFT) & 0xFu; }

// ================================================================================================
// start of the tick: the scheduled ground-truth changes, one block
// ================================================================================================
__global__ __launch_bounds__(BLOCK) void sp_begin_kernel(DevState s, uint32_t t, const FaultRec* faults, uint32_t nfaults) {
  __shared__ unsigned long long evd_sh;
  __shared__ unsigned dropped_sh;
  if (threadIdx.x == 0) { evd_sh = 0; dropped_sh = 0; }
  __syncthreads();
  const uint32_t cur = t & 1u;
  for (uint32_t k0 = threadIdx.x; k0 < nfaults; k0 += blockDim.x) {
    if (k0 && faults[k0 - 1].member == faults[k0].member) continue;      // not the first change of its member
    unsigned long long evd = 0; unsigned dropped = 0;
    for (uint32_t k = k0; k < nfaults && faults[k].member == faults[k0].member; ++k) {
      const uint32_t mbr = faults[k].member, up = faults[k].up;
      const uint32_t b = s.mb[mbr];
      if (sb_up(b) == up) continue;
      s.first_suspect[mbr] = NONE32;
      if (!up) {                                    // the process is gone, its piggyback queue with it
        s.crash_tick[mbr] = t;
        s.mb[mbr] = 0;
        continue;
      }
      if (!is_local(s, mbr)) { s.mb[mbr] = (uint8_t)(MB_UP | (1u << MB_PBN_SHIFT)); continue; }   // its owner does the rest (and publishes it)
      // (re)join: new incarnation, announce Alive: the queue holds exactly that rumour
      const uint32_t ml = mbr - s.lo;
      const uint2 hot = s.hot[ml];
      uint32_t ni = hot.x + 1;
      if (ni > INC_MAX) { atomicOr(&s.g[G_ERR], (uint32_t)ERRF_INC); ni = INC_MAX; }
      evd += h4(TAG_INC, ((uint64_t)t << 32) | mbr, ni, 0);
      const uint32_t akey = (ni << 2) | ST_ALIVE;
      uint2* line = sp_line(s, cur, ml);
      line[0] = make_uint2(mbr, pe_hi(akey, s.L));
      for (int q = 1; q < PB_SLOTS; ++q) line[q] = make_uint2(0u, 0u);
      s.mb[mbr] = (uint8_t)(MB_UP | (1u << MB_PBN_SHIFT));
      s.hot[ml] = make_uint2(ni, hot.y);
      if (s.event_mask & (1u << 4)) {
        const uint32_t pos = atomicAdd(&s.g[G_EVCUR], 1u);
        if (pos < s.event_cap) s.events[pos] = make_uint4(t, mbr, mbr, (akey << 8) | 4u);
        else dropped++;
      }
    }
    if (evd) atomicAdd(&evd_sh, evd);
    if (dropped) atomicAdd(&dropped_sh, dropped);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (evd_sh) s.blk[(size_t)s.nblocks * C_COUNT + C_EVDIGEST] += evd_sh;
    if (dropped_sh) s.blk[(size_t)s.nblocks * C_COUNT + C_EVENTS_DROPPED] += dropped_sh;
  }
}

// ================================================================================================
// probe kernel: one period of failureDetector / probeNode' per member (src/Core.hs:233-269), one WAVE per member
// ================================================================================================
// Everything a lane computes here is wave-uniform (same member, same hashes) except two things: its slice of the member's
// map (MT entries: "is c Alive in my view" is a compare per lane and a ballot) and its share of the period's deliveries.
template <int MT>
struct SpView {
  uint32_t subj[MT], key[MT];
};
template <int MT>
__device__ inline bool sp_alive(const SpView<MT>& v, uint32_t c) {
  bool hit = false;
#pragma unroll
  for (int m = 0; m < MT; ++m) hit |= v.subj[m] == c && (v.key[m] & 3u) != ST_ALIVE;
  return __ballot(hit) == 0ull;                      // no entry, or an Alive one: `isAlive` (src/Core.hs:33-34)
}
// kRandomMembers (src/Core.hs:69-74) + shuffle (src/Util.hs:37-42): the draws of swim_device.h's select_members.  The hashes
// of eight picks x eight attempts are evaluated by the LANES (one draw each: a wave in which every lane computes the same
// hash spends its time in quarter-rate multiplies -- 4 400 of the kernel's 5 000 us at 2 M members, profiles/r04c_*), the
// sequential part -- a draw is rejected if it is the member itself, excluded, picked before or not Alive in the member's
// view -- consumes them in order through lane reads.  Pick p ends up in lane p's `mine`; returns the number of picks.
template <int MT>
__device__ inline uint32_t sp_select(const DevState& s, const SpView<MT>& v, uint32_t mk, uint32_t i, uint32_t n, uint32_t purpose,
                                     uint32_t hi_idx, uint32_t excl, uint32_t lane, uint32_t& mine) {
  uint32_t np = 0;
  const uint32_t N = s.NT;
  mine = NONE32;
  auto eligible = [&](uint32_t cand) -> bool {       // wave-uniform argument and result
    if (cand == i || cand == excl) return false;     // D15; D7: the target is no proxy of itself
    if (__ballot(lane < np && mine == cand)) return false;
    return sp_alive<MT>(v, cand);
  };
  for (uint32_t p0 = 0; p0 < n; p0 += 8u) {
    const uint32_t pl = p0 + (lane >> 3);
    const uint32_t base = (purpose << 24) | (purpose == P_SELECT ? (pl << 8) : ((hi_idx << 16) | (pl << 8)));
    const uint32_t draw = __umulhi(hash_mk(mk, base | (lane & 7u), 0), N);      // lane = pick * 8 + attempt
    for (uint32_t p = p0; p < n && p < p0 + 8u; ++p) {
      uint32_t c = 0;
      bool found = false;
      for (uint32_t a = 0; a < SEL_ATTEMPTS; ++a) {
        c = (uint32_t)__builtin_amdgcn_readlane((int)draw, (int)((p - p0) * 8u + a));
        if (eligible(c)) { found = true; break; }
      }
      if (!found) {                                  // fewer candidates than draws hit: the cyclic scan (test/Spec.hs:117-128)
        const uint32_t cs = (c + 1 == N) ? 0 : c + 1;
        for (uint32_t d = 0; d < N; ++d) {
          c = cs + d; if (c >= N) c -= N;
          if (eligible(c)) { found = true; break; }
        }
      }
      if (!found) return np;
      if (lane == np) mine = c;
      ++np;
    }
  }
  return np;
}

// (Measured and dropped, profiles/r04g_*: drawing the proxies of EVERY probe ahead of the outcomes, so that the targets' and all
// proxies' bytes travel in one round of gathers instead of one round per failed probe -- 3.9 ms against 3.2 ms per launch at
// 2 M members: the kernel is bound by the instructions of the selection, not by those round trips.)
template <int MT>
__global__ __launch_bounds__(BLOCK) void sp_probe_kernel(DevState s, uint32_t t, uint32_t tk) {
  const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  const uint32_t nwaves = gridDim.x * SP_WAVES;
  // per-lane counters, summed over the wave's members, reduced once at the end
  unsigned c_pings = 0, c_active = 0, c_payloads = 0, c_rumors = 0, c_dfail = 0, c_preqs = 0, c_susp = 0, c_fsusp = 0;
  __shared__ BlockCounters sh;
  ctr_init(&sh);
  SECT_BEGIN(32);
  for (uint32_t li = blockIdx.x * SP_WAVES + wv; li < s.N; li += nwaves) {
    const uint32_t i = s.lo + li;                    // global id (hashes, targets, sources); li indexes what this handle owns
    // one round of loads: the member's byte, its map's length and the map itself (whatever its length: the loads do not wait for it)
    const uint32_t myb = s.mb[i];
    SpView<MT> v;
    {
      const uint32_t n = s.sp_tab_n[li];
      const uint32_t* rs = sp_row(s, li, 0); const uint32_t* rk = sp_row(s, li, 1);
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const uint32_t e = lane + 64u * (uint32_t)m;
        const uint32_t sj = e < s.C ? rs[e] : NONE32, ky = e < s.C ? rk[e] : 0u;
        v.subj[m] = e < n ? sj : NONE32;
        v.key[m] = e < n ? ky : 0u;
      }
    }
    if (!sb_up(myb)) continue;                       // wave-uniform
    const uint32_t mk = mix32(tk ^ i);
    SECT(32);                                         // own byte + map
    // the period's deliveries "dst merges src's start-of-tick queue", dealt to the lanes: delivery number x goes to lane
    // x mod 64; a full round is flushed -- every lane appends its delivery to its destination's inbox, the atomics of a round
    // in flight together.  Deliveries to the member itself (Acks, relayed Acks) go to its own list without atomics.
    uint32_t nd = 0, my_dst = NONE32, my_src = 0;     // my_*: this lane's delivery of the current round
    uint32_t nack = 0, nfail = 0;
    // a destination on another shard: the delivery leaves as an 8-byte record {dst, src} in one of 64 unsorted lists (a
    // counter each: one list would serialise the waves on one word), routed to its owner by sp_route_kernel
    auto flush = [&]() {
      const bool remote = my_dst != NONE32 && !is_local(s, my_dst);
      if (my_dst != NONE32 && !remote) push(s, t, my_dst - s.lo, my_src);
      if (s.n_shards > 1) {
        const unsigned long long rb = __ballot(remote);
        if (rb) {
          const uint32_t list = blockIdx.x & 63u;
          uint32_t base = 0;
          if (lane == 0) base = atomicAdd(&s.sp_ord_n[list * 16u], (uint32_t)__popcll(rb));
          base = (uint32_t)__builtin_amdgcn_readlane((int)base, 0);
          if (remote) {
            const uint32_t pos = base + (uint32_t)__popcll(rb & ((1ull << lane) - 1ull));
            if (pos < s.sp_ord_cap) s.sp_ord[(size_t)list * s.sp_ord_cap + pos] = make_uint2(my_dst, my_src);
            else atomicOr(&s.g[G_ERR], (uint32_t)ERRF_XCHG);
          }
        }
      }
      my_dst = NONE32;
    };
    auto deliver = [&](uint32_t dst, uint32_t src, uint32_t srcb) {    // wave-uniform arguments
      const uint32_t cnt = sb_qn(srcb);
      if (!cnt) return;                               // empty payload
      if (lane == 0) { c_payloads++; c_rumors += cnt; }
      if (dst == i) {
        if (lane == 0) s.ackfrom[(size_t)li * s.sp_ack_cap + nack] = src;
        nack++;
        return;
      }
      if (lane == (nd & 63u)) { my_dst = dst; my_src = src; }
      nd++;
      if ((nd & 63u) == 0u) flush();
    };
    // ms <- kRandomMembers store (numToGossip cfg) []        (src/Core.hs:239): target p in lane p, its byte gathered by lane p
    uint32_t pick = NONE32;
    const uint32_t np = sp_select<MT>(s, v, mk, i, s.P, P_SELECT, 0, NONE32, lane, pick);
    const uint32_t pickb = lane < np ? s.mb[pick] : 0u;
    if (lane == 0) { c_pings += np; c_active++; }
    SECT(33);                                         // target selection (+ their bytes issued)
    // Direct (Ping seq j) is delivered iff not lost and j is up (src/Core.hs:246); j answers Ack (:97-99), which may be lost
    // too: lane 2p evaluates the Ping's loss hash, lane 2p + 1 the Ack's
    unsigned long long lostm;
    {
      const uint32_t p = lane >> 1;
      const uint32_t j = __shfl(pick, (int)p, 64);
      const bool l = lane < 2u * np && ((lane & 1u) ? lost(s, tk, P_L_ACK, j, i, p) : lost(s, tk, P_L_PING, i, j, p));
      lostm = __ballot(l);
    }
    const unsigned long long upm = __ballot(lane < np && sb_up(pickb));
    for (uint32_t p = 0; p < np; ++p) {
      const uint32_t j = (uint32_t)__builtin_amdgcn_readlane((int)pick, (int)p), bj = (uint32_t)__builtin_amdgcn_readlane((int)pickb, (int)p);
      const bool ping_ok = ((upm >> p) & 1ull) && !((lostm >> (2u * p)) & 1ull);
      const bool ack_ok = ping_ok && !((lostm >> (2u * p + 1u)) & 1ull);
      if (ping_ok) deliver(j, i, myb);
      if (ack_ok) { deliver(i, j, bj); continue; }
      // unlessAck (D2, D3): K proxies, not the target (src/Core.hs:249; D7): proxy k in lane k, its byte gathered by lane k
      if (lane == 0) c_dfail++;
      uint32_t q_ = NONE32;
      const uint32_t nq = sp_select<MT>(s, v, mk, i, s.K, P_PROXY, p, j, lane, q_);
      const uint32_t qb_ = lane < nq ? s.mb[q_] : 0u;
      if (lane == 0) c_preqs += nq;
      // the chains i -> q -> j -> q -> i: lane 4k + h evaluates the loss hash of hop h of proxy k
      unsigned long long lm;
      {
        const uint32_t k = lane >> 2, h = lane & 3u, idx = (p << 8) | k;
        const uint32_t q = __shfl(q_, (int)k, 64);
        bool l = false;
        if (lane < 4u * nq)
          l = h == 0u ? lost(s, tk, P_L_REQ, i, q, idx) : h == 1u ? lost(s, tk, P_L_FWD, q, j, idx)
            : h == 2u ? lost(s, tk, P_L_BACK, j, q, idx) : lost(s, tk, P_L_RELAY, q, i, idx);
        lm = __ballot(l);
      }
      const unsigned long long qup = __ballot(lane < nq && sb_up(qb_));
      const bool jup = sb_up(bj) != 0u;
      bool acked = false;
      for (uint32_t k = 0; k < nq; ++k) {
        const uint32_t q = (uint32_t)__builtin_amdgcn_readlane((int)q_, (int)k), bq = (uint32_t)__builtin_amdgcn_readlane((int)qb_, (int)k);
        const uint32_t lk = (uint32_t)(lm >> (4u * k)) & 15u;
        // i -> q : IndirectPing (src/Core.hs:250, 262-269)
        if ((lk & 1u) || !((qup >> k) & 1ull)) continue;
        deliver(q, i, myb);
        // q -> j : Ping on behalf of i (src/Core.hs:105-108; D8, D12)
        if (!jup || (lk & 2u)) continue;
        deliver(j, q, bq);
        // j -> q : Ack
        if (lk & 4u) continue;
        deliver(q, j, bj);
        // q -> i : relayed Ack (D9)
        if (lk & 8u) continue;
        deliver(i, q, bq);
        acked = true;
      }
      if (!acked) {                                  // second unlessAck (src/Core.hs:251): suspectNode (:253) lands in merge
        if (lane == 0) {
          s.fail[(size_t)li * s.P + nfail] = j;
          c_susp++;
          if (jup) c_fsusp++;
          else atomicMin(&s.first_suspect[j], t);
        }
        nfail++;
      }
    }
    SECT(34);                                         // outcomes, proxies, chains
    flush();
    if (lane == 0) s.sp_out[li] = np | (nfail << 5) | (nack << 10);
    SECT(35);                                         // inbox appends
  }
  ctr_add_wave(&sh, C_PINGS, c_pings);
  ctr_add_wave(&sh, C_ACTIVE, c_active);
  ctr_add_wave(&sh, C_PAYLOADS, c_payloads);
  ctr_add_wave(&sh, C_RUMORS_SEEN, c_rumors);
  ctr_add_wave(&sh, C_DIRECT_FAILED, c_dfail);
  ctr_add_wave(&sh, C_PING_REQS, c_preqs);
  ctr_add_wave(&sh, C_SUSPECTS, c_susp);
  ctr_add_wave(&sh, C_FALSE_SUSPECTS, c_fsusp);
  ctr_flush(s, &sh, blockIdx.x);
}


// ------------------------------------------------------------------------------------------------
// The same period with one member per LANE (the default; sp_probe_kernel above stays for A/B: SWIMSIM_SP_PROBE=wave).  With a
// wave per member nearly every instruction of the period is wave-uniform -- 64 lanes compute one member's hashes and tests -- and
// the kernel was bound by exactly those instructions (3.2 ms per launch at 2 M members, profiles/r04h_*).  What made the wave
// necessary was "is c Alive in my view": a search of the member's map.  But a draw is a uniform pick among ALL members and the map
// holds <= C of them: the answer is almost always "no entry".  So sp_merge_kernel leaves a filter of 16 bits per map entry next to
// every member's map (sp_bloom: the non-Alive subjects, one hash) and a lane tests ONE BIT per draw (a 4-byte load from the member's
// own 128-byte filter: staging the tile's filters in LDS cost occupancy and lost, 1 340 against 1 160 us); for the few set bits (6 %
// at a full map, nearly all false positives) the WAVE looks the subject up in that member's map -- 64 entries per step, one ballot.  Control flow is wave-uniform throughout (loops over probes, proxies and
// attempts run while any lane needs them, lanes are predicated): every __ballot sees the whole wave.
// Draws, hashes, loss, deliveries: sp_probe_kernel's, line by line; only the order of a member's inbox / Ack-list entries differs
// (nothing depends on it: the end of the tick is stated over sets).
template <int PMAX>
__global__ __launch_bounds__(BLOCK) void sp_probe_lane_kernel(DevState s, uint32_t t, uint32_t tk) {
  __shared__ BlockCounters sh;
  __shared__ uint32_t pick_sh[PMAX][BLOCK];          // this period's targets / the current probe's proxies, per lane
  __shared__ uint32_t prox_sh[PMAX][BLOCK];
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  unsigned c_pings = 0, c_active = 0, c_payloads = 0, c_rumors = 0, c_dfail = 0, c_preqs = 0, c_susp = 0, c_fsusp = 0;
  ctr_init(&sh);
  SECT_BEGIN(32);
  const uint32_t N = s.NT;
  for (uint32_t base = blockIdx.x * BLOCK; base < s.N; base += gridDim.x * BLOCK) {
    const uint32_t li = base + tid;
    const bool valid = li < s.N;
    const uint32_t i = s.lo + li;
    const uint32_t myb = valid ? s.mb[i] : 0u;
    const bool act = valid && sb_up(myb);
    const uint32_t mk = mix32(tk ^ i);
    SECT(32);                                         // own byte
    // "c is not Alive in li's view": the filter bit first ...
    auto filter_hit = [&](uint32_t c) -> bool {
      const uint32_t b = sp_bloom_bit(s, c);
      return valid && ((s.sp_bloom[(size_t)li * sp_bloom_words(s) + (b >> 5)] >> (b & 31u)) & 1u);
    };
    // ... then, for the lanes whose bit is set, the wave walks that member's map: 64 entries per step (called by the whole wave)
    auto confirm = [&](bool need, uint32_t c) -> bool {
      bool dead = false;
      for (unsigned long long m = __ballot(need); m; m &= m - 1ull) {
        const int L = __ffsll((unsigned long long)m) - 1;
        const uint32_t liL = (uint32_t)__builtin_amdgcn_readlane((int)li, L), cL = (uint32_t)__builtin_amdgcn_readlane((int)c, L);
        const uint32_t nL = s.sp_tab_n[liL];
        const uint32_t* rs = sp_row(s, liL, 0); const uint32_t* rk = sp_row(s, liL, 1);
        bool hit = false;
        for (uint32_t e0 = 0; e0 < nL; e0 += 64u) {
          const uint32_t e = e0 + lane;
          if (e < nL && rs[e] == cL) hit = (rk[e] & 3u) != ST_ALIVE;
        }
        const bool any = __ballot(hit) != 0ull;
        if ((int)lane == L) dead = any;
      }
      return dead;
    };
    // one lane alone (the cyclic scan behind eight failed draws: tiny clusters only)
    auto dead_scan = [&](uint32_t c) -> bool {
      if (!filter_hit(c)) return false;
      const uint32_t n = s.sp_tab_n[li];
      const uint32_t* rs = sp_row(s, li, 0); const uint32_t* rk = sp_row(s, li, 1);
      for (uint32_t e = 0; e < n; ++e) if (rs[e] == c) return (rk[e] & 3u) != ST_ALIVE;
      return false;
    };
    // kRandomMembers (src/Core.hs:69-74): pick number p of a selection -- the draws of sp_select, one lane one member.  `want`: this lane
    // is selecting; earlier picks of the selection in tab[0..cnt); returns found (the pick in c).
    auto select_one = [&](bool want, uint32_t (*tab)[BLOCK], uint32_t cnt, uint32_t p, uint32_t purpose, uint32_t hi_idx, uint32_t excl, uint32_t& c) -> bool {
      const uint32_t hb = (purpose << 24) | (purpose == P_SELECT ? (p << 8) : ((hi_idx << 16) | (p << 8)));
      bool found = false;
      c = 0;
      auto fresh = [&](uint32_t d) -> bool {
        if (d == i || d == excl) return false;        // D15; D7: the target is no proxy of itself
        for (uint32_t q = 0; q < cnt; ++q) if (tab[q][tid] == d) return false;
        return true;
      };
      for (uint32_t a = 0; a < (uint32_t)SEL_ATTEMPTS; ++a) {
        const bool trying = want && !found;
        if (!__ballot(trying)) break;
        const uint32_t d = __umulhi(hash_mk(mk, hb | a, 0), N);
        if (trying) c = d;
        const bool ok = trying && fresh(d);
        const bool dead = confirm(ok && filter_hit(d), d);
        if (ok && !dead) found = true;
      }
      if (want && !found) {                           // fewer candidates than draws hit: the cyclic scan (test/Spec.hs:117-128)
        const uint32_t cs = (c + 1 == N) ? 0 : c + 1;
        for (uint32_t d = 0; d < N; ++d) {
          uint32_t x = cs + d; if (x >= N) x -= N;
          if (fresh(x) && !dead_scan(x)) { c = x; found = true; break; }
        }
      }
      return found;
    };
    uint32_t nack = 0, nfail = 0;
    uint32_t ncall = blockIdx.x * SP_WAVES + (tid >> 6);   // (wave-uniform) which of the 64 remote lists the next call appends to
    // "dst merges src's start-of-tick queue" (called by the whole wave, `on` = this lane has one)
    auto deliver = [&](bool on, uint32_t dst, uint32_t src, uint32_t srcb) {
      const uint32_t cnt = sb_qn(srcb);
      on = on && cnt != 0u;                           // empty payload: nothing travels
      if (on) { c_payloads++; c_rumors += cnt; }
      const bool self = on && dst == i;
      if (self) { s.ackfrom[(size_t)li * s.sp_ack_cap + nack] = src; nack++; }
      const bool local = on && !self && is_local(s, dst);
      if (local) push(s, t, dst - s.lo, src);
      if (s.n_shards > 1) {
        // a destination on another shard: an 8-byte record {dst, src} in one of 64 unsorted lists, one atomic per wave and call; the
        // calls of a wave go round the lists (their capacity is a 64th of a tick's records + slack: a small shard has few waves)
        const bool remote = on && !self && !local;
        const unsigned long long rb = __ballot(remote);
        if (rb) {
          const uint32_t list = ncall++ & 63u;
          const int L0 = __ffsll((unsigned long long)rb) - 1;
          uint32_t b0 = 0;
          if ((int)lane == L0) b0 = atomicAdd(&s.sp_ord_n[list * 16u], (uint32_t)__popcll(rb));
          b0 = (uint32_t)__builtin_amdgcn_readlane((int)b0, L0);
          if (remote) {
            const uint32_t pos = b0 + (uint32_t)__popcll(rb & ((1ull << lane) - 1ull));
            if (pos < s.sp_ord_cap) s.sp_ord[(size_t)list * s.sp_ord_cap + pos] = make_uint2(dst, src);
            else atomicOr(&s.g[G_ERR], (uint32_t)ERRF_XCHG);
          }
        }
      }
    };
    // ms <- kRandomMembers store (numToGossip cfg) []        (src/Core.hs:239)
    uint32_t np = 0;
    {
      bool going = act;
      for (uint32_t p = 0; p < s.P; ++p) {
        if (!__ballot(going)) break;
        uint32_t c;
        const bool f = select_one(going, pick_sh, np, p, P_SELECT, 0u, NONE32, c);
        if (going && f) { pick_sh[np][tid] = c; np++; }
        else going = false;                           // no candidate left: the selection ends (`return np`)
      }
    }
    if (act) { c_pings += np; c_active++; }
    SECT(33);                                         // target selection
    for (uint32_t p = 0; p < s.P; ++p) {
      const bool on = act && p < np;
      if (!__ballot(on)) break;
      const uint32_t j = on ? pick_sh[p][tid] : 0u;
      const uint32_t bj = on ? s.mb[j] : 0u;
      // Direct (Ping seq j) is delivered iff not lost and j is up (src/Core.hs:246); j answers Ack (:97-99), which may be lost too
      const bool ping_ok = on && sb_up(bj) && !lost(s, tk, P_L_PING, i, j, p);
      const bool ack_ok = ping_ok && !lost(s, tk, P_L_ACK, j, i, p);
      deliver(ping_ok, j, i, myb);
      deliver(ack_ok, i, j, bj);
      // unlessAck (D2, D3): K proxies, not the target (src/Core.hs:249; D7)
      const bool failed = on && !ack_ok;
      if (!__ballot(failed)) continue;
      if (failed) c_dfail++;
      const bool jup = sb_up(bj) != 0u;
      bool acked = false, going = failed;
      uint32_t nq = 0;
      for (uint32_t k = 0; k < s.K; ++k) {
        if (!__ballot(going)) break;
        uint32_t q;
        const bool f = select_one(going, prox_sh, nq, k, P_PROXY, p, j, q);
        const bool have = going && f;
        if (!have) going = false;
        // the chain i -> q -> j -> q -> i of proxy number nq (src/Core.hs:250, 262-269, 105-108; D8, D9, D12)
        const uint32_t idx = (p << 8) | nq;
        const uint32_t bq = have ? s.mb[q] : 0u;
        const bool h0 = have && sb_up(bq) && !lost(s, tk, P_L_REQ, i, q, idx);
        const bool h1 = h0 && jup && !lost(s, tk, P_L_FWD, q, j, idx);
        const bool h2 = h1 && !lost(s, tk, P_L_BACK, j, q, idx);
        const bool h3 = h2 && !lost(s, tk, P_L_RELAY, q, i, idx);
        deliver(h0, q, i, myb);
        deliver(h1, j, q, bq);
        deliver(h2, q, j, bj);
        deliver(h3, i, q, bq);
        acked |= h3;
        if (have) { prox_sh[nq][tid] = q; nq++; }
      }
      if (failed) c_preqs += nq;
      if (failed && !acked) {                         // second unlessAck (src/Core.hs:251): suspectNode (:253) lands in merge
        s.fail[(size_t)li * s.P + nfail] = j;
        nfail++;
        c_susp++;
        if (jup) c_fsusp++;
        else atomicMin(&s.first_suspect[j], t);
      }
    }
    SECT(34);                                         // outcomes, proxies, chains, inbox appends
    if (act) s.sp_out[li] = np | (nfail << 5) | (nack << 10);
    SECT(35);                                         // (the picks in LDS are per-lane columns: nothing to wait for between tiles)                                   // the tile's filters and picks are done with
  }
  ctr_add_wave(&sh, C_PINGS, c_pings);
  ctr_add_wave(&sh, C_ACTIVE, c_active);
  ctr_add_wave(&sh, C_PAYLOADS, c_payloads);
  ctr_add_wave(&sh, C_RUMORS_SEEN, c_rumors);
  ctr_add_wave(&sh, C_DIRECT_FAILED, c_dfail);
  ctr_add_wave(&sh, C_PING_REQS, c_preqs);
  ctr_add_wave(&sh, C_SUSPECTS, c_susp);
  ctr_add_wave(&sh, C_FALSE_SUSPECTS, c_fsusp);
  ctr_flush(s, &sh, blockIdx.x);
}

// ================================================================================================
// merge kernel: one member's end of tick per WAVE, its map as a hash table in LDS
// ================================================================================================
template <uint32_t CPHYS>
struct SpTable {
  uint32_t hs[CPHYS];       // subject, NONE32 = free
  uint32_t hk[CPHYS];       // (key << 2) | who stated it first: the atomicMax target of every proposal
  uint32_t h0[CPHYS];       // the entry's key at the start of the tick (0: no entry = the default) | SP_KEPT
  uint32_t newl[CPHYS];     // slots of the subjects that got an entry in this tick, in the order their creators came
  uint32_t newr[CPHYS];     // ... and their ranks (filled when the capacity needs them)
  uint32_t hist[256];       // radix select; then (write-back) the member's next "not Alive in my view" filter, <= SP_BLOOM_WORDS_MAX words
  uint32_t cl[CPHYS / 4u + 8u];            // slots of the entries that changed and stayed (<= C <= CPHYS / 4)
  uint32_t cs[CPHYS / 4u + 8u];            // ... and their subjects
  uint32_t srcs[128];       // the first 64 own-Ack sources and the first 64 inbox sources of the member, staged for the lanes
  uint32_t nnew, refute1, full, sel_b, sel_need, sel_cnt;
  uint2 qnew[PB_SLOTS];     // the head of the next queue line: this tick's rumours, by subject
};
template <uint32_t CPHYS>
__device__ inline uint32_t sp_hash(uint32_t subject) {
  uint32_t bits = 0;
  while ((1u << bits) < CPHYS) ++bits;
  return (subject * 0x9E3779B1u) >> (32u - bits);
}
// slot of `subject`, NONE32 if it has none
template <uint32_t CPHYS>
__device__ inline uint32_t sp_find(const SpTable<CPHYS>& T, uint32_t subject) {
  uint32_t h = sp_hash<CPHYS>(subject);
  for (uint32_t probe = 0; probe < CPHYS; ++probe) {
    const uint32_t cur = T.hs[h];
    if (cur == subject) return h;
    if (cur == NONE32) return NONE32;
    h = (h + 1u) & (CPHYS - 1u);
  }
  return NONE32;
}
// The state rule on one proposal: entry := max(entry, (incarnation, state)) -- suspectOrDeadNode' (src/Core.hs:142-187) and
// the unwritten aliveNode (:197-218, D6) as the commutative merge (H3, D13), here literally one LDS atomicMax.  A subject
// without an entry gets one if the proposal beats the default (key 0 = Alive@0); one that does not leaves nothing.
// `floor_rank` (0 in all but the rarest ticks): a subject WITHOUT an entry whose rank is below it is not taken in -- see the
// retry loop of sp_merge_kernel.
template <uint32_t CPHYS>
__device__ inline void sp_propose(SpTable<CPHYS>& T, uint32_t subject, uint32_t key, uint32_t prio, uint32_t mk, uint32_t floor_rank) {
  uint32_t h = sp_hash<CPHYS>(subject);
  for (uint32_t probe = 0; probe < CPHYS; ++probe) {
    uint32_t cur = T.hs[h];
    if (cur == NONE32) {
      if (key == 0u) return;
      if (floor_rank && mix32(subject ^ mk) < floor_rank) return;
      cur = atomicCAS(&T.hs[h], NONE32, subject);    // claimed by whoever comes first; another lane may hold the same rumour
      if (cur == NONE32) { cur = subject; T.newl[atomicAdd(&T.nnew, 1u)] = h; }   // the creator lists the new entry
    }
    if (cur == subject) { atomicMax(&T.hk[h], (key << 2) | prio); return; }
    h = (h + 1u) & (CPHYS - 1u);
  }
  T.full = 1u;                                        // more subjects in one tick than the working set holds: the caller retries with a floor
}

// position of a lane's item among the items of the lanes below it, and their total (wave-uniform call)
__device__ inline uint32_t sp_rank_of(bool flag, uint32_t lane, uint32_t* total) {
  const unsigned long long b = __ballot(flag);
  *total = (uint32_t)__popcll(b);
  return (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
}

// WAVES members per workgroup: the tables of a workgroup must fit its LDS allocation
template <uint32_t CPHYS, int MT, uint32_t WAVES>
__global__ __launch_bounds__(64 * WAVES) void sp_merge_kernel(DevState s, uint32_t t, uint32_t tk) {
  __shared__ SpTable<CPHYS> tabs[WAVES];
  __shared__ BlockCounters sh;
  constexpr uint32_t SPL = CPHYS / 64u;              // slots per lane
  const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  const uint32_t nwaves = gridDim.x * WAVES;
  SpTable<CPHYS>& T = tabs[wv];
  unsigned c_changes = 0, c_timers = 0, c_fdead = 0, c_refutes = 0, c_evicted = 0, c_pbw = 0, c_evdrop = 0;
  unsigned long long evd = 0;
  SECT_BEGIN(0);
  ctr_init(&sh);
  if (blockIdx.x == 0 && threadIdx.x == 0) s.g[G_OVF0 + ((t + 1) & 1u)] = 0;   // next tick's overflow list
  for (uint32_t k = 0; k < SPL; ++k) { const uint32_t x = lane + 64u * k; T.hs[x] = NONE32; T.hk[x] = 0; T.h0[x] = 0; }
  if (lane == 0) { T.full = 0; T.nnew = 0; T.refute1 = 0; }
  lds_wave_sync();
  const uint32_t cur = t & 1u;
  for (uint32_t li = blockIdx.x * WAVES + wv; li < s.N; li += nwaves) {
    const uint32_t i = s.lo + li;                    // global id
    // ---- ONE round of loads for everything the member's end of tick reads that depends on nothing else: its byte, counts,
    // the whole map (C entries whatever its length), the first 64 sources of either kind, its failed probes, its queue line.
    // (One member after the other with every load waiting for the one before was 20 round trips per member-tick.)
    const uint32_t myb = s.mb[i];
    const uint2 hot0 = s.hot[li];
    const uint32_t po = s.sp_out[li];
    const uint32_t cnt = s.inbox_cnt[li];
    const uint32_t n0 = s.sp_tab_n[li];
    uint32_t msub[MT], mkey[MT], msince[MT];
    {
      const uint32_t* rs = sp_row(s, li, 0); const uint32_t* rk = sp_row(s, li, 1); const uint32_t* rt = sp_row(s, li, 2);
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const uint32_t e = lane + 64u * (uint32_t)m;
        msub[m] = e < s.C ? rs[e] : NONE32; mkey[m] = e < s.C ? rk[e] : 0u; msince[m] = e < s.C ? rt[e] : 0u;
      }
    }
    const uint32_t w_ack = lane < s.sp_ack_cap ? s.ackfrom[(size_t)li * s.sp_ack_cap + lane] : 0u;
    const uint32_t w_in = lane < s.inbox_cap ? s.inbox[(size_t)li * s.inbox_cap + lane] : 0u;
    const uint32_t w_fail = lane < s.P ? s.fail[(size_t)li * s.P + lane] : 0u;
    uint2 oe = make_uint2(0u, 0u);
    if (lane < (uint32_t)PB_SLOTS) oe = sp_line(s, cur, li)[lane];
    if (!sb_up(myb)) continue;                       // wave-uniform: a member that is down does nothing and receives nothing
    const uint32_t mk = mix32(tk ^ i);
    const uint32_t nsent = po & 31u, nfail = (po >> 5) & 31u, nack = po >> 10;
    T.srcs[lane] = w_ack; T.srcs[64u + lane] = w_in;
    SECT(0);                                          // the member's inputs (one round of loads)
    // this lane's entries of the map: which exist, whose suspicion deadline is due (the FIXME at src/Core.hs:141; D4)
    bool mhave[MT], mdue[MT];
    uint32_t mslot[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      mhave[m] = lane + 64u * (uint32_t)m < n0;
      mdue[m] = mhave[m] && (mkey[m] & 3u) == ST_SUSPECT && msince[m] - 1u + s.S <= t;
      mslot[m] = 0;
    }
    // The working set of a tick -- the map and every subject the member hears of for the first time -- has CPHYS >= 4 C
    // slots.  When a tick brings more (a member with hundreds of sources: large P and K under heavy loss, a degraded
    // cluster), subjects without an entry are taken in only above a rank FLOOR: they would be the first to go anyway -- an
    // entry that appears in this tick stays only if its rank is among the C largest of the tick's changes -- PROVIDED at
    // least C of the tick's changes lie at or above the floor, which is checked; the floor is found by bisection (every
    // step a full pass over the member's inputs: the rare path buys exactness, not speed).
    uint32_t floor_rank = 0, floor_lo = 0, floor_hi = 0xFFFFFFFFu;
    uint32_t nnew = 0, nchold = 0;                    // entries that appeared; old entries that changed
    bool mch[MT];
    for (;;) {
      // ---- the map into the hash table
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        if (!mhave[m]) continue;
        uint32_t h = sp_hash<CPHYS>(msub[m]);
        for (;;) {                                     // subjects of a map are distinct: a free slot is mine
          if (atomicCAS(&T.hs[h], NONE32, msub[m]) == NONE32) break;
          h = (h + 1u) & (CPHYS - 1u);
        }
        T.hk[h] = mdue[m] ? ((((mkey[m] & ~3u) | ST_DEAD) << 2) | SP_PRIO_TIMER) : (mkey[m] << 2);
        T.h0[h] = mkey[m];
        mslot[m] = h;
      }
      lds_wave_sync();
      SECT(1);                                        // map -> hash table
      // ---- own probes that ended without any Ack: Suspect at the incarnation the map holds (src/Core.hs:253)
      if (lane < nfail) {
        const uint32_t j = w_fail;
        const uint32_t sl = sp_find<CPHYS>(T, j);
        const uint32_t k0 = sl == NONE32 ? 0u : (T.h0[sl] & 0xFFFFFFu);
        sp_propose<CPHYS>(T, j, (k0 & ~3u) | ST_SUSPECT, SP_PRIO_PROBE, mk, floor_rank);
      }
      // ---- the rumours received this tick (src/Core.hs:110-117): 8 sources x 8 queue entries per round of loads
      {
        const uint32_t nin = cnt < s.inbox_cap ? cnt : s.inbox_cap;
        const uint32_t nsrc = nack + nin;
        auto entry = [&](uint2 e) {
          if (!pe_tx(e.y)) return;
          const uint32_t key = pe_key(e.y);
          if (e.x == i) {                             // about self -> refute (:155-166); incarnations below my own are stale (:151)
            if ((key & 3u) != ST_ALIVE && (key >> 2) >= hot0.x) atomicMax(&T.refute1, (key >> 2) + 1u);
            return;
          }
          sp_propose<CPHYS>(T, e.x, key, SP_PRIO_GOSSIP, mk, floor_rank);
        };
        // source x of the member: its own Ack sources first, then its inbox (the first 64 of either are staged in LDS)
        auto src_of = [&](uint32_t x) -> uint32_t {
          if (x < nack) return x < 64u ? T.srcs[x] : s.ackfrom[(size_t)li * s.sp_ack_cap + x];
          const uint32_t y = x - nack;
          return y < 64u ? T.srcs[64u + y] : s.inbox[(size_t)li * s.inbox_cap + y];
        };
        // four rounds (32 sources) of queue entries in flight at a time
        for (uint32_t x0 = 0; x0 < nsrc; x0 += 32u) {
          uint2 en[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const uint32_t x = x0 + 8u * (uint32_t)r + (lane >> 3);
            en[r] = make_uint2(0u, 0u);
            if (x < nsrc) en[r] = sp_src_line(s, cur, src_of(x))[lane & 7u];
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) entry(en[r]);
        }
        if (cnt > s.inbox_cap) {                      // the exact overflow list (rare): my entries of it, one source at a time
          const uint32_t novf = min(s.g[G_OVF0 + (t & 1u)], s.ovf_cap);
          for (uint32_t y0 = 0; y0 < novf; y0 += 64u) {
            const uint32_t y = y0 + lane;
            uint2 o = make_uint2(NONE32, 0u);
            if (y < novf) o = s.ovf[(size_t)(t & 1u) * s.ovf_cap + y];
            unsigned long long hits = __ballot(o.x == li);
            for (; hits; hits &= hits - 1ull) {
              const int L = __ffsll((unsigned long long)hits) - 1;
              const uint32_t src = (uint32_t)__builtin_amdgcn_readlane((int)o.y, L);
              if (lane < 8u) entry(sp_src_line(s, cur, src)[lane]);
            }
          }
        }
      }
      lds_wave_sync();
      SECT(2);                                        // failed probes + received rumours (lines, proposals)
      // ---- what changed: this lane's entries of the map (the entries that appeared are listed in newl)
      nnew = T.nnew;
      uint32_t myabove = 0;                           // this lane's share of the tick's changes at or above the floor
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        mch[m] = mhave[m] && (T.hk[mslot[m]] >> 2) > mkey[m];
        myabove += (mch[m] && (!floor_rank || mix32(msub[m] ^ mk) >= floor_rank)) ? 1u : 0u;
      }
      const bool full = T.full != 0u;
      bool ok = !full;
      if (ok && floor_rank) ok = wave_sum(myabove) + nnew >= s.C;   // (an entry that appeared lies above the floor by construction)
      if (ok) break;                                  // wave-uniform
      // the working set overflowed (raise the floor) or the floor cut into the C that stay (lower it): once more
      if (full) floor_lo = floor_rank; else floor_hi = floor_rank;
      floor_rank = floor_lo + (floor_hi - floor_lo) / 2u;
      if (floor_rank == floor_lo) floor_rank = floor_lo + 1u;
      lds_wave_sync();
      for (uint32_t k = 0; k < SPL; ++k) { const uint32_t x = lane + 64u * k; T.hs[x] = NONE32; T.hk[x] = 0; T.h0[x] = 0; }
      if (lane == 0) { T.full = 0; T.nnew = 0; }
      lds_wave_sync();
    }
    {
      uint32_t mine = 0;
#pragma unroll
      for (int m = 0; m < MT; ++m) mine += mch[m] ? 1u : 0u;
      nchold = MT == 1 ? (uint32_t)__popcll(__ballot(mch[0])) : wave_sum(mine);
    }
    const uint32_t total = n0 + nnew, nchanged = nchold + nnew;
    SECT(3);                                          // what changed, how many
    // ---- the capacity: the C entries with the largest (lastChange, rank) stay; rank = mix32(subject ^ mk) is a keyed
    // permutation of the ids (no ties).  Everything that changed in this tick has the same, the largest, lastChange: so either
    // the changes fit (A: all of them stay, the C - changes most recent of the others with them) or they do not (B: the C
    // changes of the largest rank stay, nothing else).  Radix select of the threshold in LDS, a byte per pass, over the
    // candidates only: in case A the untouched entries of the map, which sit in the lanes' registers.
    const bool evicting = total > s.C;
    const bool caseB = nchanged > s.C;
    unsigned long long thr = 0ull;                   // a candidate stays iff its priority >= thr
    uint32_t mrank[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) mrank[m] = 0;
    if (evicting) {
#pragma unroll
      for (int m = 0; m < MT; ++m) mrank[m] = mhave[m] ? mix32(msub[m] ^ mk) : 0u;
      if (caseB) for (uint32_t x = lane; x < nnew; x += 64u) T.newr[x] = mix32(T.hs[T.newl[x]] ^ mk);
      uint32_t need = caseB ? s.C : s.C - nchanged;  // candidates that stay
      // a lane's candidates: case A the untouched entries it holds; case B the changed ones it holds and its share of the new ones
      auto for_candidates = [&](auto&& f) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
          if (mhave[m] && mch[m] == caseB) f(((unsigned long long)(caseB ? 0u : msince[m]) << 32) | mrank[m]);
        if (caseB) for (uint32_t x = lane; x < nnew; x += 64u) f((unsigned long long)T.newr[x]);
      };
      if (need == 0u) thr = ~0ull;                    // (case A with exactly C changes: no untouched entry stays)
      else {
        // the bytes of lastChange every candidate shares need no pass (case B: all of them)
        uint32_t smin = NONE32, smax = 0;
        if (!caseB) {
#pragma unroll
          for (int m = 0; m < MT; ++m) if (mhave[m] && !mch[m]) { smin = min(smin, msince[m]); smax = max(smax, msince[m]); }
          smin = NONE32 - wave_max_all(NONE32 - smin); smax = wave_max_all(smax);
        } else { smin = 0; smax = 0; }
        lds_wave_sync();
        int pass = 7;
        unsigned long long prefix = 0ull;             // the decided high bytes of the threshold
        while (pass >= 4 && (smin >> (8 * (pass - 4))) == (smax >> (8 * (pass - 4)))) { prefix = smin >> (8 * (pass - 4)); pass--; }
        bool done = false;
        for (; pass >= 0 && !done; --pass) {
          for (uint32_t b = lane; b < 256u; b += 64u) T.hist[b] = 0;
          lds_wave_sync();
          for_candidates([&](unsigned long long p) {
            if (pass == 7 || (p >> (8 * (pass + 1))) == prefix) atomicAdd(&T.hist[(uint32_t)(p >> (8 * pass)) & 0xFFu], 1u);
          });
          lds_wave_sync();
          // the bin in which the need-th largest lies: lane l owns bins 4l .. 4l+3
          const uint32_t b0 = T.hist[4u * lane], b1 = T.hist[4u * lane + 1u], b2 = T.hist[4u * lane + 2u], b3 = T.hist[4u * lane + 3u];
          const uint32_t mine = b0 + b1 + b2 + b3;
          const uint32_t incl = wave_prefix_incl(mine);
          const uint32_t all = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
          const uint32_t above = all - incl;           // candidates in the bins of higher lanes
          if (above < need && need <= above + mine) {  // exactly one lane
            uint32_t a = above, b = 4u * lane + 3u, hb = b3;
            if (a + b3 < need) { a += b3; b = 4u * lane + 2u; hb = b2;
              if (a + b2 < need) { a += b2; b = 4u * lane + 1u; hb = b1;
                if (a + b1 < need) { a += b1; b = 4u * lane; hb = b0; } } }
            T.sel_b = b; T.sel_need = need - a; T.sel_cnt = hb;
          }
          lds_wave_sync();
          prefix = (prefix << 8) | T.sel_b;
          need = T.sel_need;
          if (T.sel_cnt == need) { thr = prefix << (8 * pass); done = true; }   // the whole bin stays
        }
        if (!done) thr = prefix;
      }
    }
    SECT(4);                                          // radix select
    // ---- who stays; the entries that changed AND stayed are listed (slot, subject), the map's first, then the new ones
    bool mkeep[MT];
    uint32_t ncl = 0, nkept_old = 0;
    {
      uint32_t base_cl = 0, base_keep = 0;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const unsigned long long p = ((unsigned long long)((caseB || mch[m]) ? 0u : msince[m]) << 32) | mrank[m];
        mkeep[m] = mhave[m] && (!evicting || (mch[m] ? (!caseB || p >= thr) : (!caseB && p >= thr)));
        if (mhave[m] && !mkeep[m]) c_evicted++;       // an entry of the start of the tick that leaves
        uint32_t tot = 0;
        const uint32_t r = sp_rank_of(mkeep[m] && mch[m], lane, &tot);
        if (mkeep[m] && mch[m]) { T.cl[base_cl + r] = mslot[m]; T.cs[base_cl + r] = msub[m]; }
        base_cl += tot;
        uint32_t totk = 0;
        (void)sp_rank_of(mkeep[m], lane, &totk);
        base_keep += totk;
        if (mkeep[m]) T.h0[mslot[m]] |= SP_KEPT;
      }
      nkept_old = base_keep;
      for (uint32_t x0 = 0; x0 < nnew; x0 += 64u) {
        const uint32_t x = x0 + lane;
        bool keep = false; uint32_t sl = 0;
        if (x < nnew) { sl = T.newl[x]; keep = !caseB || (unsigned long long)T.newr[x] >= thr; }
        uint32_t tot = 0;
        const uint32_t r = sp_rank_of(keep, lane, &tot);
        if (keep) { T.cl[base_cl + r] = sl; T.cs[base_cl + r] = T.hs[sl]; T.h0[sl] |= SP_KEPT; }
        base_cl += tot;
      }
      ncl = base_cl;
    }
    lds_wave_sync();
    SECT(5);                                          // who stays
    // ---- account for what changed and stayed (`saveMember m'`, src/Core.hs:169-179): digest, counters, events
    const unsigned long long ha = mix64(mix64((uint64_t)TAG_EV) + (((uint64_t)t << 32) | i));
#pragma unroll
    for (int m = 0; m < MT; ++m)                      // `deadNode` after the timeout, for the deadlines whose entry stayed
      if (mdue[m] && mkeep[m]) { c_timers++; c_fdead += sb_up(s.mb[msub[m]]) ? 1u : 0u; }
    for (uint32_t c0 = 0; c0 < ncl; c0 += 64u) {
      const uint32_t c = c0 + lane;
      bool ev = false; uint32_t subject = 0, key = 0, cause = 0;
      if (c < ncl) {
        const uint32_t x = T.cl[c];
        subject = T.cs[c]; key = T.hk[x] >> 2;
        const uint32_t pr = T.hk[x] & 3u, k0 = T.h0[x] & 0xFFFFFFu;
        cause = pr == SP_PRIO_TIMER ? 1u : pr == SP_PRIO_PROBE ? 0u : 2u;
        evd += (mix64(ha + subject) | 1ull) * (unsigned long long)(key - k0);
        c_changes++;
        ev = (s.event_mask & (1u << cause)) != 0u;
      }
      const unsigned long long evb = __ballot(ev);
      if (evb) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&s.g[G_EVCUR], (uint32_t)__popcll(evb));
        base = (uint32_t)__builtin_amdgcn_readlane((int)base, 0);
        if (ev) {
          const uint32_t pos = base + (uint32_t)__popcll(evb & ((1ull << lane) - 1ull));
          if (pos < s.event_cap) s.events[pos] = make_uint4(t, i, subject, (key << 8) | cause);
          else c_evdrop++;
        }
      }
    }
    SECT(6);                                          // accounting: digest, counters, events
    // ---- refutation: bump own incarnation past the rumour's (src/Core.hs:155-166; D10)
    uint32_t self_inc = hot0.x;
    const uint32_t refute1 = T.refute1;
    const bool refuted = refute1 != 0u;
    uint32_t akey = 0;
    if (refuted) {
      uint32_t ni = refute1;                          // = the largest non-Alive incarnation about me + 1
      if (ni > INC_MAX) { atomicOr(&s.g[G_ERR], (uint32_t)ERRF_INC); ni = INC_MAX; }
      self_inc = ni;
      akey = (ni << 2) | ST_ALIVE;
      if (lane == 0) {
        evd += h4(TAG_INC, ((uint64_t)t << 32) | i, ni, 0);
        c_refutes++;
        if (s.event_mask & (1u << 3)) {
          const uint32_t pos = atomicAdd(&s.g[G_EVCUR], 1u);
          if (pos < s.event_cap) s.events[pos] = make_uint4(t, i, i, (akey << 8) | 3u /*REFUTE*/);
          else c_evdrop++;
        }
        T.cs[ncl] = i;                                // the refutation is one more rumour of the tick
      }
    }
    // ---- the queue (D5): this tick's rumours -- what changed and stayed, the refutation -- with a full budget, by subject;
    // then the aged survivors in their order; the 8 best.  Every rumour of the tick counts the subjects below its own: the
    // 8 smallest take the head of the line in that order.
    const uint32_t ncand = ncl + (refuted ? 1u : 0u);
    const uint32_t gn = min(ncand, (uint32_t)PB_SLOTS);
    lds_wave_sync();
    for (uint32_t c = lane; c < ncand; c += 64u) {
      const uint32_t sj = T.cs[c];
      uint32_t below = 0;
      for (uint32_t o = 0; o < ncand; ++o) below += T.cs[o] < sj ? 1u : 0u;
      if (below < (uint32_t)PB_SLOTS) T.qnew[below] = make_uint2(sj, pe_hi(c < ncl ? (T.hk[T.cl[c]] >> 2) : akey, s.L));
    }
    lds_wave_sync();
    SECT(7);                                          // the tick's rumours by subject
    const uint32_t age = nsent ? nsent : 1u;
    const uint32_t oldn = sb_qn(myb);
    uint32_t nout = gn;
    {
      // lanes 0..7 look at one old entry each; it survives if its budget lasts and no rumour of this tick supersedes it
      bool keep = false;
      if (lane < oldn) {
        const uint32_t tx = pe_tx(oe.y);
        bool superseded = refuted && oe.x == i;
        if (!superseded && oe.x != i) {
          const uint32_t sl = sp_find<CPHYS>(T, oe.x);
          superseded = sl != NONE32 && (T.h0[sl] & SP_KEPT) && (T.hk[sl] >> 2) > (T.h0[sl] & 0xFFFFFFu);
        }
        keep = tx > age && !superseded;
        oe.y = pe_hi(pe_key(oe.y), tx - age);
      }
      const unsigned long long kb = __ballot(keep);
      const uint32_t rank = (uint32_t)__popcll(kb & ((1ull << lane) - 1ull));
      uint2* out = sp_line(s, cur ^ 1u, li);
      if (keep && gn + rank < (uint32_t)PB_SLOTS) out[gn + rank] = oe;
      nout = min((uint32_t)PB_SLOTS, gn + (uint32_t)__popcll(kb));
      if (lane < gn) out[lane] = T.qnew[lane];
      if (lane >= nout && lane < (uint32_t)PB_SLOTS) out[lane] = make_uint2(0u, 0u);
    }
    if (lane == 0) {
      c_pbw += (oldn || nout) ? 1u : 0u;
      s.mb[i] = (uint8_t)(MB_UP | (nout << MB_PBN_SHIFT));
      if (self_inc != hot0.x) s.hot[li] = make_uint2(self_inc, hot0.y);
      s.inbox_cnt[li] = 0;
    }
    SECT(8);                                          // queue line + member state stores
    // ---- the map back to HBM: the entries of the map that stay, then the new ones that do; the slots in use cleared
    {
      uint32_t* rs = sp_row(s, li, 0); uint32_t* rk = sp_row(s, li, 1); uint32_t* rt = sp_row(s, li, 2);
      uint32_t base = 0;
      const uint32_t bw = sp_bloom_words(s);
      for (uint32_t w = lane; w < bw; w += 64u) T.hist[w] = 0u;
      lds_wave_sync();
      auto bloom_put = [&](uint32_t subject, uint32_t key) {    // sp_probe_lane_kernel's filter: the subjects that are not Alive here
        if ((key & 3u) != ST_ALIVE) { const uint32_t b = sp_bloom_bit(s, subject); atomicOr(&T.hist[b >> 5], 1u << (b & 31u)); }
      };
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        uint32_t tot = 0;
        const uint32_t r = sp_rank_of(mkeep[m], lane, &tot);
        if (mkeep[m]) {
          const uint32_t kf = T.hk[mslot[m]] >> 2;
          rs[base + r] = msub[m]; rk[base + r] = kf; rt[base + r] = mch[m] ? t + 1u : msince[m];
          bloom_put(msub[m], kf);
        }
        base += tot;
      }
      for (uint32_t x0 = 0; x0 < nnew; x0 += 64u) {
        const uint32_t x = x0 + lane;
        bool keep = false; uint32_t sl = 0;
        if (x < nnew) { sl = T.newl[x]; keep = (T.h0[sl] & SP_KEPT) != 0u; }
        uint32_t tot = 0;
        const uint32_t r = sp_rank_of(keep, lane, &tot);
        if (keep) { rs[base + r] = T.hs[sl]; rk[base + r] = T.hk[sl] >> 2; rt[base + r] = t + 1u; bloom_put(T.hs[sl], T.hk[sl] >> 2); }
        base += tot;
      }
      if (lane == 0) s.sp_tab_n[li] = base;
      lds_wave_sync();
      for (uint32_t w = lane; w < bw; w += 64u) s.sp_bloom[(size_t)li * bw + w] = T.hist[w];
      (void)nkept_old;
      lds_wave_sync();                                // every lane has read what it needs of the table
#pragma unroll
      for (int m = 0; m < MT; ++m) if (mhave[m]) { T.hs[mslot[m]] = NONE32; T.hk[mslot[m]] = 0; T.h0[mslot[m]] = 0; }
      for (uint32_t x = lane; x < nnew; x += 64u) { const uint32_t sl = T.newl[x]; T.hs[sl] = NONE32; T.hk[sl] = 0; T.h0[sl] = 0; }
      if (lane == 0) { T.full = 0; T.nnew = 0; T.refute1 = 0; }
    }
    lds_wave_sync();
    SECT(9);                                          // map write-back, table clear
  }
  ctr_add_wave(&sh, C_CHANGES, c_changes);
  ctr_add_wave(&sh, C_PB_WRITES, c_pbw);
  ctr_add_wave(&sh, C_TIMERS_FIRED, c_timers);
  ctr_add_wave(&sh, C_FALSE_DEADS, c_fdead);
  ctr_add_wave(&sh, C_REFUTES, c_refutes);
  ctr_add_wave(&sh, C_EVENTS_DROPPED, c_evdrop);
  ctr_add_wave(&sh, C_EVICTED, c_evicted);
  {
    const unsigned long long wevd = wave_sum64(evd);
    if (lane == 0u && wevd) atomicAdd(&sh.evd, wevd);
  }
  ctr_flush(s, &sh, blockIdx.x);
}

// ================================================================================================
// sharded clusters of bounded handles: one exchange of 8-byte delivery records per tick + an all-gather
// ================================================================================================
// Members shard by contiguous id range as on dense handles.  What a delivery "dst merges src's queue" needs from another shard is
// (a) the receiver learning of it: an 8-byte record {dst, src} to dst's owner, which appends src to dst's inbox; (b) src's
// start-of-tick queue line, 64 bytes: every shard holds a replica of everybody's line (sp_qall) and byte (mb), all-gathered at
// the start of the tick -- after the scheduled changes, so that the bytes carry this tick's ground truth.  Probe outcomes need
// nothing else (loss is a hash, selection reads the prober's own map).
__global__ __launch_bounds__(BLOCK) void sp_publish_kernel(DevState s, uint32_t t) {
  const uint32_t x = blockIdx.x * BLOCK + threadIdx.x;              // one 8-byte entry per thread: coalesced copy
  if (x < s.N * (uint32_t)PB_SLOTS) s.sp_qall[(size_t)s.lo * PB_SLOTS + x] = s.sp_q[(size_t)(t & 1u) * s.N * PB_SLOTS + x];
  if (x < 64u) s.sp_ord_n[x * 16u] = 0;                              // this tick's lists of remote deliveries
  if (x < 3u * MAX_SHARDS) s.send_cnt[x] = 0;
}
// the tick's remote deliveries (64 unsorted lists) -> per-owner segments of 16-byte records {dst, src, -, -} (p_send): a block
// counts its chunk per peer in LDS, reserves with one atomic per peer, writes
__global__ __launch_bounds__(BLOCK) void sp_route_kernel(DevState s) {
  __shared__ uint32_t cnt[MAX_SHARDS], base[MAX_SHARDS];
  for (uint32_t list = 0; list < 64u; ++list) {
    const uint32_t n = min(s.sp_ord_n[list * 16u], s.sp_ord_cap);
    const uint32_t chunk = (n + gridDim.x - 1) / gridDim.x;
    const uint32_t first = min(n, blockIdx.x * chunk), m = min(n - first, chunk);
    if (!m) continue;                                // (block-uniform)
    __syncthreads();
    if (threadIdx.x < MAX_SHARDS) cnt[threadIdx.x] = 0;
    __syncthreads();
    const uint2* ord = s.sp_ord + (size_t)list * s.sp_ord_cap + first;
    for (uint32_t k = threadIdx.x; k < m; k += BLOCK) atomicAdd(&cnt[owner_of(s, ord[k].x)], 1u);
    __syncthreads();
    if (threadIdx.x < MAX_SHARDS) {
      const uint32_t c = cnt[threadIdx.x];
      if (c) base[threadIdx.x] = atomicAdd(&s.send_cnt[1 * MAX_SHARDS + threadIdx.x], c);
      cnt[threadIdx.x] = 0;
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < m; k += BLOCK) {
      const uint2 o = ord[k];
      const uint32_t peer = owner_of(s, o.x);
      const uint32_t pos = base[peer] + atomicAdd(&cnt[peer], 1u);
      if (pos < s.p_cap) s.p_send[(size_t)peer * s.p_cap + pos] = make_uint4(o.x, o.y, 0u, 0u);
      else atomicOr(&s.g[G_ERR], (uint32_t)ERRF_XCHG);
    }
  }
}
// the records the peers sent: append src to my member's inbox.  dev_counts: the number of records of every peer's segment comes
// from sp_pin[] in device memory (swimsim_cluster_step copies the senders' counters there: no host in the loop)
__global__ __launch_bounds__(BLOCK) void sp_ingest_kernel(DevState s, uint32_t t, PeerCounts p_counts, uint32_t dev_counts) {
  for (uint32_t peer = 0; peer < s.n_shards; ++peer) {
    const uint32_t np = peer == s.shard ? 0u : min(dev_counts ? s.sp_pin[peer] : p_counts.v[peer], s.p_cap);
    for (uint32_t k = blockIdx.x * BLOCK + threadIdx.x; k < np; k += gridDim.x * BLOCK) {
      const uint4 r = s.p_recv[(size_t)peer * s.p_cap + k];
      if (is_local(s, r.x)) push(s, t, r.x - s.lo, r.y);
    }
  }
}

// ================================================================================================
// observables
// ================================================================================================
__global__ __launch_bounds__(BLOCK) void sp_digest_kernel(DevState s, uint32_t t, unsigned long long* out) {
  __shared__ unsigned long long acc;
  if (threadIdx.x == 0) acc = 0;
  __syncthreads();
  const uint32_t li = blockIdx.x * BLOCK + threadIdx.x;
  if (li < s.N) {
    const uint32_t i = s.lo + li, b = s.mb[i];
    unsigned long long mh = h4(TAG_SELF, i, s.hot[li].x, sb_up(b) ? 1u : 0u);
    const uint32_t n = s.sp_tab_n[li];
    const uint32_t* rs = sp_row(s, li, 0); const uint32_t* rk = sp_row(s, li, 1); const uint32_t* rt = sp_row(s, li, 2);
    for (uint32_t e = 0; e < n; ++e) {
      mh += h4(TAG_VIEW, rs[e], rk[e], rt[e]);
      if ((rk[e] & 3u) == ST_SUSPECT) mh += h4(TAG_TIMER, rs[e], (uint64_t)rt[e] - 1 + s.S, 0);
    }
    const uint32_t qn = sb_qn(b);
    const uint2* line = sp_line(s, t & 1u, li);        // the buffer the NEXT tick reads
    for (uint32_t q = 0; q < qn; ++q) mh += h4(TAG_PB, line[q].x, pe_key(line[q].y), pe_tx(line[q].y));
    unsigned long long d = mix64(mh + mix64((uint64_t)TAG_MEMBER + i));
    const uint32_t fs = s.first_suspect[i];
    if (fs != NONE32) d += h4(TAG_FD, i, fs, 0);
    atomicAdd(&acc, d);
  }
  __syncthreads();
  if (threadIdx.x == 0 && acc) atomicAdd(out, acc);
}

// swimsim_coverage on bounded maps: a member holds the rumour if its entry about `subject` is at least `key` (no entry = the
// default, Alive@0)
__global__ __launch_bounds__(BLOCK) void sp_coverage_kernel(DevState s, uint32_t subject, uint32_t key, unsigned long long* out) {
  const uint32_t li = blockIdx.x * BLOCK + threadIdx.x;
  uint32_t up = 0, hold = 0;
  if (li < s.N && s.lo + li != subject && sb_up(s.mb[s.lo + li])) {
    up = 1;
    uint32_t k = 0;
    const uint32_t n = s.sp_tab_n[li];
    const uint32_t* rs = sp_row(s, li, 0);
    for (uint32_t e = 0; e < n; ++e) if (rs[e] == subject) { k = sp_row(s, li, 1)[e]; break; }
    hold = k >= key ? 1u : 0u;
  }
  const unsigned long long bh = __ballot(hold != 0u), bu = __ballot(up != 0u);
  if ((threadIdx.x & 63u) == 0u) {
    if (bh) atomicAdd(&out[0], (unsigned long long)__popcll(bh));
    if (bu) atomicAdd(&out[1], (unsigned long long)__popcll(bu));
  }
}

}  // namespace swim
