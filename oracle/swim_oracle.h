/*
 * swim_oracle.h -- CPU ORACLE for the SWIM tick (TEST INFRASTRUCTURE, NOT PRODUCT).
 *
 * A plain-C, message-level restatement (sequential by default; member-range threads on request) of the reference's
 * per-member protocol rules (src/Core.hs, src/Util.hs, src/Types.hs of
 * jpfuentes2/swim) as a bulk-synchronous tick.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library.  The product
 * (libswimsim.so) never links, loads or calls it.
 *
 * PARITY PINNING: the reference cannot be built here (no GHC) and
 * its own tests pin only kRandomMembers / removeDeadNodes / Ping / IndirectPing
 * (test/Spec.hs:98-174).  Those KATs are restated in tests/test_oracle_kat.py.
 * Suspect/Dead/Alive rules, timers, dissemination and the RNG stream are NOT
 * pinned by any reference test ("parity unpinned"): for them this
 * oracle IS the specification, justified line by line
 * against src/Core.hs, with every divergence from the literal code listed in
 * DESIGN.md section 3 (defects D1-D16).
 *
 * The exported functions mirror include/swimsim.h one for one under the
 * `swimoracle_` prefix (same structs, same status codes) so that the parity
 * tests drive both through one harness; a few extra hooks expose unit-level
 * pieces (process, the literal state rule, the hash) for known-answer tests.
 */
#ifndef SWIM_ORACLE_H
#define SWIM_ORACLE_H

#include "../include/swimsim.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct swimoracle swimoracle_t;

int swimoracle_default_config(swimsim_config_t* cfg);
int swimoracle_create(const swimsim_config_t* cfg, swimoracle_t** out);
int swimoracle_create_msg(const swimsim_config_t* cfg, swimoracle_t** out, char* err, size_t errcap);
void swimoracle_destroy(swimoracle_t* h);
const char* swimoracle_last_error(const swimoracle_t* h);
int swimoracle_schedule_fault(swimoracle_t* h, uint64_t tick, uint32_t member, uint8_t up);
int swimoracle_step(swimoracle_t* h, uint32_t nticks);
int swimoracle_tick(const swimoracle_t* h, uint64_t* tick);
int swimoracle_drain_events(swimoracle_t* h, swimsim_event_t* buf, size_t cap, size_t* n_out);
int swimoracle_read_view(swimoracle_t* h, uint32_t observer, swimsim_view_entry_t* buf,
                         size_t cap, size_t* n_out);
int swimoracle_read_member(swimoracle_t* h, uint32_t member, swimsim_member_t* out);
int swimoracle_first_detect(swimoracle_t* h, uint64_t* out, size_t n);
int swimoracle_digest(swimoracle_t* h, uint64_t* out);
int swimoracle_coverage(swimoracle_t* h, uint32_t subject, uint8_t state, uint32_t incarnation, uint64_t out[2]);
int swimoracle_counters(swimoracle_t* h, uint64_t* out, size_t n);
int swimoracle_k_random_members(swimoracle_t* h, uint32_t observer, uint32_t n,
                                const uint32_t* excludes, size_t n_excludes, uint32_t* out,
                                size_t cap, size_t* n_out);
int swimoracle_set_view(swimoracle_t* h, uint32_t observer, uint32_t subject, uint8_t state,
                        uint32_t incarnation);
int swimoracle_get_config(const swimoracle_t* h, swimsim_config_t* out);
int swimoracle_inject_rumor(swimoracle_t* h, uint32_t observer, uint32_t subject, uint8_t state, uint32_t incarnation);

/* ---- oracle-only hooks ---------------------------------------------------- */

/* Message = src/Types.hs:122-145; type tags = msgIndex / MsgType (:159-178). */
enum {
  SWIMO_MSG_PING = 0,
  SWIMO_MSG_INDIRECT_PING = 1,
  SWIMO_MSG_ACK = 2,
  SWIMO_MSG_SUSPECT = 3,
  SWIMO_MSG_ALIVE = 4,
  SWIMO_MSG_DEAD = 5
};
typedef struct swimoracle_msg {
  uint8_t  type;         /* SWIMO_MSG_*                                           */
  uint32_t seq_no;       /* seqNo                                                 */
  uint32_t node;         /* node (member id)                                      */
  uint32_t target;       /* IndirectPing.target (member id stands for addr+port)  */
  uint32_t incarnation;  /* Suspect/Alive/Dead.incarnation                        */
  uint32_t dead_from;    /* Dead.deadFrom                                         */
  uint32_t to;           /* destination of a Direct gossip (output only)          */
  uint8_t  broadcast;    /* 1 = `Broadcast msg`, 0 = `Direct msg addr` (output)    */
} swimoracle_msg_t;

/* `process sender msg` for member `self` (src/Core.hs:89-117) in capture mode:
 * the resulting `[Gossip]` is written to out instead of being delivered.
 * literal_d8 != 0 reproduces the literal IndirectPing behaviour pinned by
 * test/Spec.hs:166-174 (bump storeIncarnation, use it as the Ping's seqNo);
 * the tick itself uses literal_d8 = 0 (defect D8).  State-changing messages
 * (Suspect/Dead/Alive) are applied to self's view immediately in this mode. */
int swimoracle_process(swimoracle_t* h, uint32_t self, uint32_t sender,
                       const swimoracle_msg_t* msg, int literal_d8, swimoracle_msg_t* out,
                       size_t cap, size_t* n_out);

/* The LITERAL single-message rule of suspectOrDeadNode' (src/Core.hs:142-187) on
 * packed keys (inc<<2|state): returns the entry after receiving `msg_key`
 * (state Suspect or Dead) when the current entry is `cur_key`.  Used to document
 * exactly where the oracle's commutative merge differs (D13). */
uint32_t swimoracle_reference_rule(uint32_t cur_key, uint32_t msg_key);
/* The oracle's merge: max over (incarnation, state). */
uint32_t swimoracle_merge_rule(uint32_t cur_key, uint32_t msg_key);

/* `removeDeadNodes` (src/Core.hs:65-67) on a view listing: compacts in place,
 * returns the new length. */
size_t swimoracle_remove_dead_nodes(swimsim_view_entry_t* entries, size_t n);

/* Step under the literal rule above instead of the merge (Alive messages keep the merge: aliveNode is
 * unwritten, D6).  d13_hits counts the proposals on which the two rules disagree; while it is 0 the run
 * equals the merge run in every observable. */
int swimoracle_set_literal_rule(swimoracle_t* h, int on);
uint64_t swimoracle_d13_hits(const swimoracle_t* h);

/* Spec hash H(seed, tick, a, b, c) -> u32. */
uint32_t swimoracle_hash(uint64_t seed, uint32_t tick, uint32_t a, uint32_t b, uint32_t c);

/* Split the members into n contiguous ranges stepped by n threads (the caller's thread included).
 * Every observable is independent of n; n = 1 (the default) is the plain sequential oracle.  Used by
 * bench.py's cpu_baseline ("all host cores") and by oracle-checked runs at full size. */
int swimoracle_set_threads(swimoracle_t* h, uint32_t n);

/* Permute the order in which each member applies its pending rumours / timers
 * (0 = canonical arrival order).  Results must not depend on it (property test). */
int swimoracle_set_shuffle(swimoracle_t* h, uint64_t shuffle_seed);

#ifdef __cplusplus
}
#endif
#endif
