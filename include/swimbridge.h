/*
 * swimbridge.h -- the live-node bridge: ONE UDP endpoint behind which the whole simulated
 * population answers the reference's wire protocol, so that a real `Core.main`-style node (src/Core.hs:272-287) can
 * have the simulated members as its peers.  Host-only code in libswimsim.so (POSIX sockets + the wire codec of
 * swimwire.h + the public entry points of swimsim.h); nothing of it runs on the GPU.
 *
 * It restates `handleUDPMessage.process` (src/Core.hs:79-117) for datagrams that arrive from OUTSIDE, with the
 * simulated members as the receivers.  Every datagram is an `Envelope` (src/Types.hs:96-119; the codec sends
 * Envelopes in both directions: D11).  Per message, `sender` = the datagram's source address:
 *
 *   Ping seq node          node = "m<id>", id a simulated member that is up:  `Direct (Ack seq []) sender`
 *                          (src/Core.hs:97-99) -- as a compound Envelope that also carries the member's piggyback
 *                          queue as Suspect / Alive / Dead messages (D5; row a18).  Anybody else: nothing (:100-101).
 *   IndirectPing seq t p n n = "m<j>" simulated: a simulated proxy pings it inside the simulation and relays the
 *                          answer (D9): `Ack seq []` to the sender iff m<j> is up.  Otherwise `Ping seq n` goes to
 *                          the address (t, p) (src/Core.hs:105-108; the requester's seqNo, not the reference's
 *                          incarnation counter: D8) and the first `Ack seq` that comes back from there is relayed.
 *   Ack seq _              relayed to the requester of a pending IndirectPing with that seqNo, else counted.
 *   Suspect / Alive / Dead about "m<s>": handed to the simulated member the datagram addresses -- the member its
 *                          Ping names, else member (s + 1) mod N -- through swimsim_inject_rumor: delivered in the next
 *                          tick stepped, ruled on like any other rumour (src/Core.hs:110-117).  About a name that is
 *                          not a simulated member: counted, dropped (the simulator has no entry for outsiders).
 *   undecodable datagram   counted, dropped (the reference's receiver dies: D16).
 *
 * The bridge neither steps the simulation nor owns it: the embedder alternates swimsim_step and swimbridge_poll on
 * one thread (the handle is not re-entrant).  "Real-node demo unverified": no GHC in this image -- the test peer
 * (tests/test_bridge.py) is a Python restatement of the reference's send side over the same codec.
 */
#ifndef SWIMBRIDGE_H
#define SWIMBRIDGE_H

#include <stddef.h>
#include <stdint.h>

#include "swimsim.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct swimbridge swimbridge_t;

typedef struct swimbridge_stats {
  uint64_t datagrams_in, datagrams_out, decode_errors;
  uint64_t pings, pings_unanswered;        /* Ping for a simulated member / for a member that is down or nobody  */
  uint64_t indirect_pings, relayed_acks;   /* IndirectPing handled / Acks relayed for them                       */
  uint64_t acks_in;                        /* Acks nobody was waiting for                                        */
  uint64_t rumors_injected, rumors_foreign;/* Suspect/Alive/Dead handed to the simulation / about unknown names  */
  uint64_t rumors_dropped;                 /* ... beyond what the simulation takes before its next tick (a flood) */
  uint64_t sends_failed;                   /* sendto refused the ADDRESS a datagram named (0.0.0.0:0, broadcast, ..) */
  uint64_t bare_in;                        /* datagrams that were a bare `Message` (swimbridge_accept_bare)       */
} swimbridge_stats_t;

/* Bind a UDP socket on bind_ip:port (port 0 = any free port; the reference binds 127.0.0.1:4000, src/Core.hs:278)
 * for the members of `sim`.  SWIMSIM_ERR_INVALID (also: a sharded handle -- one endpoint answers for the whole
 * population: swimbridge_open_cluster) / SWIMSIM_ERR_DEVICE (socket errors; swimbridge_last_error). */
int swimbridge_open(swimsim_t* sim, const char* bind_ip, uint16_t port, swimbridge_t** out);
/* The same endpoint for a SHARDED cluster whose shards are all handles of this process (round 6): shards[k] = shard k of n_shards of
 * one population.  A Ping is answered from the owner of the named member; a Suspect / Alive / Dead message goes to the owner of the
 * member it addresses (swimsim_inject_rumor) and is made known to every other shard (swimsim_note_outside_rumor: its subject's view
 * row belongs to the whole cluster).  The embedder alternates swimbridge_poll with the cluster's tick (swimsim_cluster_step or the
 * phase calls) on one thread. */
int swimbridge_open_cluster(swimsim_t* const* shards, uint32_t n_shards, const char* bind_ip, uint16_t port, swimbridge_t** out);
int swimbridge_port(const swimbridge_t* b, uint16_t* port);
/* The reference's send side encodes a bare `Message`, its receive side decodes an `Envelope` (D11; src/Core.hs:133-134
 * against :84): with on = 1 the bridge also accepts bare-Message datagrams -- what a LITERAL reference node sends --
 * and keeps answering with Envelopes, which is what that node's receiver parses.  Off by default. */
int swimbridge_accept_bare(swimbridge_t* b, int on);
/* Handle the datagrams that are waiting (at most max_datagrams; waits up to timeout_ms for the first one).
 * Returns the number handled (>= 0) or a negative swimsim status.  Failures caused by what a datagram SAYS (undecodable
 * bytes, an address sendto refuses, more gossip than the simulation takes before its next tick) are counted in the
 * statistics and dropped; only local socket / device failures are returned.
 * Addresses: IndirectPing.target is the reference's HostAddress (network byte order as a word, include/swimwire.h). */
int swimbridge_poll(swimbridge_t* b, int timeout_ms, uint32_t max_datagrams);
int swimbridge_stats(const swimbridge_t* b, swimbridge_stats_t* out);
const char* swimbridge_last_error(const swimbridge_t* b);
void swimbridge_close(swimbridge_t* b);

#ifdef __cplusplus
}
#endif
#endif /* SWIMBRIDGE_H */
