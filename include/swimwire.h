/*
 * swimwire.h -- C ABI of the wire codec of jpfuentes2/swim: the `Envelope`
 * single / compound framing (src/Types.hs:88-119) around msgpack message bodies
 * (`packAeson` of the Generic-derived JSON of `Message`, src/Types.hs:122-155).
 * Exported by libswimsim.so next to the simulator; host-only code (no GPU involved).
 *
 * Framing (src/Types.hs:96-119):
 *   one message    [type u8 = msgIndex][msgpack body]
 *   n > 1 messages [CompoundMsg = 6][n u8][len u16 big-endian x n][body x n]    (bodies without type byte)
 * Body = msgpack map {"tag": <constructor name>, <record fields>...} -- what aeson's default `genericToJSON`
 * gives for a sum of records, converted to msgpack by msgpack-aeson 0.1.0.0 (integers as msgpack ints, strings
 * as msgpack str, [Word8] as an array of ints).
 *
 * What is pinned and what is not: the reference's tests hold NO golden bytes, only round trips
 * (test/Spec.hs:77-96).  The key ORDER of the reference's maps is the iteration order of an
 * unordered-containers HashMap and is not reproducible without GHC; msgpack readers do not depend on it.
 * This codec therefore WRITES keys in declaration order ("tag" first) and READS any order and any msgpack
 * integer width, which is the interoperability contract with `unpackAeson` / `packAeson`.
 *
 * Bounds (row a18; src/Core.hs:280 `sourceSocket sock 65535`): a compound envelope carries at most 255
 * messages, each body at most 65 535 bytes (u16 length), and the whole datagram must fit 65 535 bytes;
 * the encoder refuses anything beyond (SWIMSIM_ERR_CAPACITY).
 */
#ifndef SWIMWIRE_H
#define SWIMWIRE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* MsgType (src/Types.hs:159-167) */
enum {
  SWIMWIRE_PING = 0, SWIMWIRE_INDIRECT_PING = 1, SWIMWIRE_ACK = 2, SWIMWIRE_SUSPECT = 3,
  SWIMWIRE_ALIVE = 4, SWIMWIRE_DEAD = 5, SWIMWIRE_COMPOUND = 6
};
#define SWIMWIRE_MAX_MSGS 255u        /* numMsgs is a Word8 (src/Types.hs:100)                 */
#define SWIMWIRE_MAX_DATAGRAM 65535u  /* src/Core.hs:280                                        */
#define SWIMWIRE_NAME_MAX 63u         /* node / deadFrom strings this ABI carries (NUL-terminated) */
#define SWIMWIRE_PAYLOAD_MAX 255u     /* Ack.payload bytes this ABI carries                      */

/* `Message` (src/Types.hs:122-145); fields not used by a constructor are ignored / zeroed. */
typedef struct swimwire_msg {
  uint8_t  type;                         /* SWIMWIRE_PING .. SWIMWIRE_DEAD                        */
  uint8_t  payload_len;                  /* Ack.payload                                           */
  uint16_t port;                         /* IndirectPing.port, Alive.port                         */
  uint32_t seq_no;                       /* Ping / IndirectPing / Ack .seqNo                      */
  uint32_t target;                       /* IndirectPing.target                                   */
  uint32_t addr;                         /* Alive.addr                                            */
  int64_t  incarnation;                  /* Suspect / Alive / Dead .incarnation (Haskell Int)     */
  char     node[SWIMWIRE_NAME_MAX + 1];      /* .node                                            */
  char     dead_from[SWIMWIRE_NAME_MAX + 1]; /* Dead.deadFrom                                    */
  uint8_t  payload[SWIMWIRE_PAYLOAD_MAX];
} swimwire_msg_t;

/* `encode (Envelope msgs)` (src/Types.hs:96-103).  n = 1 -> single form, n > 1 -> compound form.
 * Returns SWIMSIM_OK, SWIMSIM_ERR_INVALID (n = 0, bad type, name too long), SWIMSIM_ERR_CAPACITY (more than
 * 255 messages, a body or the datagram beyond 65 535 bytes) or SWIMSIM_ERR_BUFFER (*n_out = bytes needed). */
int swimwire_encode(const swimwire_msg_t* msgs, size_t n, uint8_t* buf, size_t cap, size_t* n_out);

/* `decode :: ByteString -> Either String Envelope` (src/Types.hs:105-119), with the reference's failure
 * cases: empty input, unknown type byte, "compound message is truncated", "compound mesage with zero
 * messages", a body that does not parse as a `Message`.  SWIMSIM_ERR_INVALID + swimwire_last_error() text on
 * failure; SWIMSIM_ERR_BUFFER if cap is too small (*n_out = messages in the envelope). */
int swimwire_decode(const uint8_t* buf, size_t len, swimwire_msg_t* out, size_t cap, size_t* n_out);

/* The reference's SEND side does not frame (D11): `gossip msg addr = yield $ UDP.Message (encode msg) addr`
 * (src/Core.hs:133-134) encodes a bare `Message` -- the msgpack body alone, no type byte -- while its receive side
 * (`decode`, :84) expects an Envelope.  Two entry points for talking to such a literal node:
 * swimwire_encode_bare = `encode msg` (instance Serialize Message, src/Types.hs:151-155);
 * swimwire_decode_any accepts an Envelope OR a bare Message (*was_bare = 1): the first byte tells them apart (an
 * Envelope starts with a type byte 0..6, a bare body with a msgpack map header 0x80..0x8f / 0xde / 0xdf). */
int swimwire_encode_bare(const swimwire_msg_t* msg, uint8_t* buf, size_t cap, size_t* n_out);
int swimwire_decode_any(const uint8_t* buf, size_t len, swimwire_msg_t* out, size_t cap, size_t* n_out, int* was_bare);

/* Address convention: `IndirectPing.target` and `Alive.addr` are the `HostAddress` the reference takes out of a
 * `SockAddrInet` (src/Core.hs:264-266) -- the IPv4 address as the network library holds it, i.e. the 32-bit word whose
 * in-memory bytes are in NETWORK order (sin_addr.s_addr; on x86 127.0.0.1 is 0x0100007F).  The codec carries the word
 * as a number; whoever turns it into a socket address must reinterpret it, not htonl() it (include/swimbridge.h). */

/* Encoded size of the envelope without writing it (the byte model of a piggybacked datagram). */
int swimwire_size(const swimwire_msg_t* msgs, size_t n, size_t* n_out);

/* Message of the last failed call on this thread. */
const char* swimwire_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* SWIMWIRE_H */
