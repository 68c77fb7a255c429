// swim_wire.cpp -- the reference's wire codec behind a C ABI (include/swimwire.h): `Envelope` framing
// (src/Types.hs:88-119) around msgpack bodies (src/Types.hs:151-155).  Host-only code inside libswimsim.so;
// the simulated tick has no wire, this serialises what a simulated member WOULD send (row a18: the byte
// model of a piggybacked datagram) and parses datagrams of a live node.
#include "../../include/swimwire.h"
#include "../../include/swimsim.h"

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace {

thread_local std::string g_err;
int fail(int code, const std::string& m) { g_err = m; return code; }

// ---- msgpack writer (the subset aeson's Value needs: map, str, int, array) ------------------------
struct W {
  std::vector<uint8_t> b;
  void u8(uint8_t v) { b.push_back(v); }
  void be16(uint16_t v) { u8((uint8_t)(v >> 8)); u8((uint8_t)v); }
  void be32(uint32_t v) { be16((uint16_t)(v >> 16)); be16((uint16_t)v); }
  void be64(uint64_t v) { be32((uint32_t)(v >> 32)); be32((uint32_t)v); }
  void map(uint32_t n) { if (n < 16) u8(0x80 | n); else { u8(0xde); be16((uint16_t)n); } }
  void arr(uint32_t n) { if (n < 16) u8(0x90 | n); else if (n < 65536) { u8(0xdc); be16((uint16_t)n); } else { u8(0xdd); be32(n); } }
  void str(const char* s) {
    const size_t n = strlen(s);
    if (n < 32) u8(0xa0 | (uint8_t)n); else if (n < 256) { u8(0xd9); u8((uint8_t)n); } else { u8(0xda); be16((uint16_t)n); }
    b.insert(b.end(), s, s + n);
  }
  // smallest encoding, as the Haskell msgpack library writes ObjectInt
  void i64(int64_t v) {
    if (v >= 0) {
      const uint64_t u = (uint64_t)v;
      if (u < 128) u8((uint8_t)u);
      else if (u < 256) { u8(0xcc); u8((uint8_t)u); }
      else if (u < 65536) { u8(0xcd); be16((uint16_t)u); }
      else if (u < (1ull << 32)) { u8(0xce); be32((uint32_t)u); }
      else { u8(0xcf); be64(u); }
    } else if (v >= -32) u8((uint8_t)v);
    else if (v >= -128) { u8(0xd0); u8((uint8_t)v); }
    else if (v >= -32768) { u8(0xd1); be16((uint16_t)v); }
    else if (v >= -(1ll << 31)) { u8(0xd2); be32((uint32_t)v); }
    else { u8(0xd3); be64((uint64_t)v); }
  }
};

const char* TAGS[6] = {"Ping", "IndirectPing", "Ack", "Suspect", "Alive", "Dead"};

// `packAeson (toJSON msg)`: {"tag": constructor, fields...}, keys in declaration order (src/Types.hs:122-145)
int put_body(W& w, const swimwire_msg_t& m) {
  if (m.type > SWIMWIRE_DEAD) return fail(SWIMSIM_ERR_INVALID, "encode: not a Message constructor");
  if (memchr(m.node, 0, sizeof m.node) == nullptr || memchr(m.dead_from, 0, sizeof m.dead_from) == nullptr)
    return fail(SWIMSIM_ERR_INVALID, "encode: name is not NUL-terminated");
  switch (m.type) {
    case SWIMWIRE_PING:
      w.map(3); w.str("tag"); w.str(TAGS[0]); w.str("seqNo"); w.i64(m.seq_no); w.str("node"); w.str(m.node); break;
    case SWIMWIRE_INDIRECT_PING:
      w.map(5); w.str("tag"); w.str(TAGS[1]); w.str("seqNo"); w.i64(m.seq_no); w.str("target"); w.i64(m.target);
      w.str("port"); w.i64(m.port); w.str("node"); w.str(m.node); break;
    case SWIMWIRE_ACK:
      w.map(3); w.str("tag"); w.str(TAGS[2]); w.str("seqNo"); w.i64(m.seq_no); w.str("payload"); w.arr(m.payload_len);
      for (uint32_t k = 0; k < m.payload_len; ++k) w.i64(m.payload[k]);
      break;
    case SWIMWIRE_SUSPECT:
      w.map(3); w.str("tag"); w.str(TAGS[3]); w.str("incarnation"); w.i64(m.incarnation); w.str("node"); w.str(m.node); break;
    case SWIMWIRE_ALIVE:
      w.map(5); w.str("tag"); w.str(TAGS[4]); w.str("incarnation"); w.i64(m.incarnation); w.str("node"); w.str(m.node);
      w.str("addr"); w.i64(m.addr); w.str("port"); w.i64(m.port); break;
    default:
      w.map(4); w.str("tag"); w.str(TAGS[5]); w.str("incarnation"); w.i64(m.incarnation); w.str("node"); w.str(m.node);
      w.str("deadFrom"); w.str(m.dead_from); break;
  }
  return SWIMSIM_OK;
}

int build(const swimwire_msg_t* msgs, size_t n, std::vector<uint8_t>* out) {
  if (!msgs || n == 0) return fail(SWIMSIM_ERR_INVALID, "encode: an Envelope holds at least one message (NonEmpty)");
  if (n > SWIMWIRE_MAX_MSGS) return fail(SWIMSIM_ERR_CAPACITY, "encode: more than 255 messages in one envelope");
  out->clear();
  if (n == 1) {                                     // put (Envelope (msg :| [])) = putWord8 (msgIndex msg) >> put msg
    W w; w.u8(msgs[0].type);
    int rc = put_body(w, msgs[0]); if (rc) return rc;
    *out = std::move(w.b);
  } else {                                          // CompoundMsg, count, u16be lengths, bodies (src/Types.hs:98-103)
    std::vector<std::vector<uint8_t>> bodies(n);
    for (size_t k = 0; k < n; ++k) {
      W w; int rc = put_body(w, msgs[k]); if (rc) return rc;
      if (w.b.size() > 65535) return fail(SWIMSIM_ERR_CAPACITY, "encode: message body beyond 65 535 bytes (u16 length)");
      bodies[k] = std::move(w.b);
    }
    out->push_back(SWIMWIRE_COMPOUND); out->push_back((uint8_t)n);
    for (auto& b : bodies) { out->push_back((uint8_t)(b.size() >> 8)); out->push_back((uint8_t)b.size()); }
    for (auto& b : bodies) out->insert(out->end(), b.begin(), b.end());
  }
  if (out->size() > SWIMWIRE_MAX_DATAGRAM) return fail(SWIMSIM_ERR_CAPACITY, "encode: datagram beyond 65 535 bytes (src/Core.hs:280)");
  return SWIMSIM_OK;
}

// ---- msgpack reader ---------------------------------------------------------------------------------
struct R {
  const uint8_t* p; const uint8_t* e; bool ok = true;
  bool need(size_t n) { if ((size_t)(e - p) < n) { ok = false; return false; } return true; }
  uint8_t u8() { if (!need(1)) return 0; return *p++; }
  uint64_t be(int n) { uint64_t v = 0; if (!need((size_t)n)) return 0; for (int k = 0; k < n; ++k) v = (v << 8) | *p++; return v; }
  // any integer width (the writer on the other side chooses); floats with integral value as aeson would accept
  bool integer(int64_t* out, bool* is_u64, uint64_t* u) {
    *is_u64 = false;
    const uint8_t t = u8(); if (!ok) return false;
    if (t < 0x80) { *out = t; return true; }
    if (t >= 0xe0) { *out = (int8_t)t; return true; }
    switch (t) {
      case 0xcc: *out = (int64_t)be(1); return ok;
      case 0xcd: *out = (int64_t)be(2); return ok;
      case 0xce: *out = (int64_t)be(4); return ok;
      case 0xcf: *u = be(8); *is_u64 = true; *out = (int64_t)*u; return ok;
      case 0xd0: *out = (int8_t)be(1); return ok;
      case 0xd1: *out = (int16_t)be(2); return ok;
      case 0xd2: *out = (int32_t)be(4); return ok;
      case 0xd3: *out = (int64_t)be(8); return ok;
      default: ok = false; return false;
    }
  }
  bool str(std::string* s) {
    const uint8_t t = u8(); if (!ok) return false;
    size_t n;
    if ((t & 0xe0) == 0xa0) n = t & 31; else if (t == 0xd9) n = be(1); else if (t == 0xda) n = be(2); else if (t == 0xdb) n = be(4);
    else { ok = false; return false; }
    if (!ok || !need(n)) return false;
    s->assign(reinterpret_cast<const char*>(p), n); p += n;
    return true;
  }
  bool container(uint8_t fix, uint8_t t16, uint8_t t32, uint32_t* n) {
    const uint8_t t = u8(); if (!ok) return false;
    if ((t & 0xf0) == fix) { *n = t & 15; return true; }
    if (t == t16) { *n = (uint32_t)be(2); return ok; }
    if (t == t32) { *n = (uint32_t)be(4); return ok; }
    ok = false; return false;
  }
  // skip one value of any type (fields a newer peer might add: aeson's generic parser ignores them too)
  void skip(int depth = 0) {
    if (depth > 32) { ok = false; return; }
    const uint8_t t = u8(); if (!ok) return;
    auto adv = [&](size_t n) { if (need(n)) p += n; };
    if (t < 0x80 || t >= 0xe0 || t == 0xc0 || t == 0xc2 || t == 0xc3) return;
    if ((t & 0xe0) == 0xa0) { adv(t & 31); return; }
    if ((t & 0xf0) == 0x90) { for (uint32_t k = t & 15; k && ok; --k) skip(depth + 1); return; }
    if ((t & 0xf0) == 0x80) { for (uint32_t k = 2 * (t & 15); k && ok; --k) skip(depth + 1); return; }
    switch (t) {
      case 0xcc: case 0xd0: adv(1); return;
      case 0xcd: case 0xd1: adv(2); return;
      case 0xce: case 0xd2: case 0xca: adv(4); return;
      case 0xcf: case 0xd3: case 0xcb: adv(8); return;
      case 0xc4: case 0xd9: adv(be(1)); return;
      case 0xc5: case 0xda: adv(be(2)); return;
      case 0xc6: case 0xdb: adv(be(4)); return;
      case 0xdc: { uint32_t n = (uint32_t)be(2); for (; n && ok; --n) skip(depth + 1); return; }
      case 0xdd: { uint32_t n = (uint32_t)be(4); for (; n && ok; --n) skip(depth + 1); return; }
      case 0xde: { uint32_t n = 2 * (uint32_t)be(2); for (; n && ok; --n) skip(depth + 1); return; }
      case 0xdf: { uint64_t n = 2 * be(4); for (; n && ok; --n) skip(depth + 1); return; }
      default: ok = false; return;
    }
  }
};

// `unpackAeson` + `parseJSON` of one body: a map with "tag" and the constructor's record fields, in any order
int get_body(const uint8_t* b, size_t len, swimwire_msg_t* m) {
  R r{b, b + len};
  uint32_t n = 0;
  if (!r.container(0x80, 0xde, 0xdf, &n)) return fail(SWIMSIM_ERR_INVALID, "Could not parse message body: not a map");
  memset(m, 0, sizeof *m);
  std::string tag; bool has_tag = false;
  enum { F_SEQ = 1, F_NODE = 2, F_TARGET = 4, F_PORT = 8, F_PAYLOAD = 16, F_INC = 32, F_ADDR = 64, F_DEADFROM = 128 };
  unsigned seen = 0;
  auto bounded = [&](const char* what, int64_t lo, uint64_t hi, uint64_t* out) -> bool {
    int64_t v; bool isu; uint64_t u = 0;
    if (!r.integer(&v, &isu, &u)) { fail(SWIMSIM_ERR_INVALID, std::string("Could not parse message body: ") + what + " is not an integer"); return false; }
    if (isu ? (u > hi) : (v < lo || (v >= 0 && (uint64_t)v > hi))) { fail(SWIMSIM_ERR_INVALID, std::string("Could not parse message body: ") + what + " out of bounds"); return false; }
    *out = isu ? u : (uint64_t)v;
    return true;
  };
  auto name = [&](const char* what, char* dst) -> bool {
    std::string s;
    if (!r.str(&s)) { fail(SWIMSIM_ERR_INVALID, std::string("Could not parse message body: ") + what + " is not a string"); return false; }
    if (s.size() > SWIMWIRE_NAME_MAX || s.find('\0') != std::string::npos) { fail(SWIMSIM_ERR_CAPACITY, std::string(what) + " longer than this ABI carries"); return false; }
    memcpy(dst, s.data(), s.size()); dst[s.size()] = 0;
    return true;
  };
  for (uint32_t k = 0; k < n; ++k) {
    std::string key;
    if (!r.str(&key)) return fail(SWIMSIM_ERR_INVALID, "Could not parse message body: key is not a string");
    uint64_t v = 0;
    if (key == "tag") { if (!r.str(&tag)) return fail(SWIMSIM_ERR_INVALID, "Could not parse message body: tag is not a string"); has_tag = true; }
    else if (key == "seqNo") { if (!bounded("seqNo", 0, 0xFFFFFFFFull, &v)) return SWIMSIM_ERR_INVALID; m->seq_no = (uint32_t)v; seen |= F_SEQ; }
    else if (key == "target") { if (!bounded("target", 0, 0xFFFFFFFFull, &v)) return SWIMSIM_ERR_INVALID; m->target = (uint32_t)v; seen |= F_TARGET; }
    else if (key == "addr") { if (!bounded("addr", 0, 0xFFFFFFFFull, &v)) return SWIMSIM_ERR_INVALID; m->addr = (uint32_t)v; seen |= F_ADDR; }
    else if (key == "port") { if (!bounded("port", 0, 0xFFFFull, &v)) return SWIMSIM_ERR_INVALID; m->port = (uint16_t)v; seen |= F_PORT; }
    else if (key == "incarnation") {
      int64_t iv; bool isu; uint64_t u = 0;
      if (!r.integer(&iv, &isu, &u) || (isu && u > 0x7FFFFFFFFFFFFFFFull)) return fail(SWIMSIM_ERR_INVALID, "Could not parse message body: incarnation is not an Int");
      m->incarnation = iv; seen |= F_INC;
    }
    else if (key == "node") { if (!name("node", m->node)) return g_err.find("longer") != std::string::npos ? SWIMSIM_ERR_CAPACITY : SWIMSIM_ERR_INVALID; seen |= F_NODE; }
    else if (key == "deadFrom") { if (!name("deadFrom", m->dead_from)) return g_err.find("longer") != std::string::npos ? SWIMSIM_ERR_CAPACITY : SWIMSIM_ERR_INVALID; seen |= F_DEADFROM; }
    else if (key == "payload") {
      uint32_t pn = 0;
      if (!r.container(0x90, 0xdc, 0xdd, &pn)) return fail(SWIMSIM_ERR_INVALID, "Could not parse message body: payload is not an array");
      if (pn > SWIMWIRE_PAYLOAD_MAX) return fail(SWIMSIM_ERR_CAPACITY, "payload longer than this ABI carries");
      for (uint32_t q = 0; q < pn; ++q) { if (!bounded("payload byte", 0, 255, &v)) return SWIMSIM_ERR_INVALID; m->payload[q] = (uint8_t)v; }
      m->payload_len = (uint8_t)pn; seen |= F_PAYLOAD;
    }
    else r.skip();
    if (!r.ok) return fail(SWIMSIM_ERR_INVALID, "Could not parse message body: truncated");
  }
  if (r.p != r.e) return fail(SWIMSIM_ERR_INVALID, "Could not parse message body: trailing bytes");
  if (!has_tag) return fail(SWIMSIM_ERR_INVALID, "Could not parse message body: no tag");
  static const unsigned NEED[6] = {F_SEQ | F_NODE, F_SEQ | F_TARGET | F_PORT | F_NODE, F_SEQ | F_PAYLOAD, F_INC | F_NODE,
                                   F_INC | F_NODE | F_ADDR | F_PORT, F_INC | F_NODE | F_DEADFROM};
  for (int t = 0; t < 6; ++t)
    if (tag == TAGS[t]) {
      if ((seen & NEED[t]) != NEED[t]) return fail(SWIMSIM_ERR_INVALID, "Could not parse message body: a field of " + tag + " is missing");
      m->type = (uint8_t)t;
      return SWIMSIM_OK;
    }
  return fail(SWIMSIM_ERR_INVALID, "Could not parse message body: unknown constructor " + tag);
}

}  // namespace

extern "C" {

const char* swimwire_last_error(void) { return g_err.c_str(); }

int swimwire_size(const swimwire_msg_t* msgs, size_t n, size_t* n_out) {
  if (!n_out) return SWIMSIM_ERR_INVALID;
  std::vector<uint8_t> b;
  int rc = build(msgs, n, &b);
  *n_out = b.size();
  return rc;
}

int swimwire_encode(const swimwire_msg_t* msgs, size_t n, uint8_t* buf, size_t cap, size_t* n_out) {
  if (!n_out) return SWIMSIM_ERR_INVALID;
  std::vector<uint8_t> b;
  int rc = build(msgs, n, &b);
  if (rc) { *n_out = 0; return rc; }
  *n_out = b.size();
  if (b.size() > cap || !buf) return SWIMSIM_ERR_BUFFER;
  memcpy(buf, b.data(), b.size());
  return SWIMSIM_OK;
}

// `encode msg` of a bare `Message` (instance Serialize Message, src/Types.hs:151-155): the msgpack body alone, no type byte --
// what the reference's send side literally puts on the wire (src/Core.hs:133-134; D11)
int swimwire_encode_bare(const swimwire_msg_t* msg, uint8_t* buf, size_t cap, size_t* n_out) {
  if (!n_out || !msg) return SWIMSIM_ERR_INVALID;
  W w;
  const int rc = put_body(w, *msg);
  if (rc) { *n_out = 0; return rc; }
  if (w.b.size() > SWIMWIRE_MAX_DATAGRAM) return fail(SWIMSIM_ERR_CAPACITY, "encode: datagram beyond 65 535 bytes (src/Core.hs:280)");
  *n_out = w.b.size();
  if (w.b.size() > cap || !buf) return SWIMSIM_ERR_BUFFER;
  memcpy(buf, w.b.data(), w.b.size());
  return SWIMSIM_OK;
}

// An Envelope, or the bare `Message` of the literal sender.  The two cannot be confused: an Envelope starts with a type
// byte 0..6, a bare Message with the header of a msgpack map (0x80..0x8f, 0xde, 0xdf).
int swimwire_decode_any(const uint8_t* buf, size_t len, swimwire_msg_t* out, size_t cap, size_t* n_out, int* was_bare) {
  if (was_bare) *was_bare = 0;
  if (!n_out || (!buf && len)) return SWIMSIM_ERR_INVALID;
  if (len && ((buf[0] & 0xf0) == 0x80 || buf[0] == 0xde || buf[0] == 0xdf)) {
    if (was_bare) *was_bare = 1;
    *n_out = 0;
    if (cap < 1 || !out) { *n_out = 1; return SWIMSIM_ERR_BUFFER; }          // (room for one message is what it needs)
    const int rc = get_body(buf, len, &out[0]);
    if (rc == SWIMSIM_OK) *n_out = 1;
    return rc;
  }
  return swimwire_decode(buf, len, out, cap, n_out);
}

int swimwire_decode(const uint8_t* buf, size_t len, swimwire_msg_t* out, size_t cap, size_t* n_out) {
  if (!n_out || (!buf && len)) return SWIMSIM_ERR_INVALID;
  *n_out = 0;
  if (len == 0) return fail(SWIMSIM_ERR_INVALID, "too few bytes");
  const uint8_t typ = buf[0];
  if (typ > SWIMWIRE_COMPOUND) return fail(SWIMSIM_ERR_INVALID, "invalid message type " + std::to_string(typ));
  if (typ != SWIMWIRE_COMPOUND) {                   // _ -> Envelope . (:| []) <$> get : the type byte is not looked at again
    *n_out = 1;
    if (cap < 1 || !out) return SWIMSIM_ERR_BUFFER;
    return get_body(buf + 1, len - 1, &out[0]);
  }
  if (len < 2) return fail(SWIMSIM_ERR_INVALID, "too few bytes");
  const size_t n = buf[1];
  if (len - 2 < n * 2) return fail(SWIMSIM_ERR_INVALID, "compound message is truncated");
  if (n == 0) return fail(SWIMSIM_ERR_INVALID, "compound mesage with zero messages");
  *n_out = n;
  if (cap < n || !out) return SWIMSIM_ERR_BUFFER;
  size_t off = 2 + 2 * n;
  for (size_t k = 0; k < n; ++k) {
    const size_t bl = ((size_t)buf[2 + 2 * k] << 8) | buf[3 + 2 * k];
    if (len - off < bl) return fail(SWIMSIM_ERR_INVALID, "too few bytes (isolate)");
    int rc = get_body(buf + off, bl, &out[k]);   // `isolate len get`: the body must use exactly its bytes
    if (rc) return rc;
    off += bl;
  }
  return SWIMSIM_OK;
}

}  // extern "C"
