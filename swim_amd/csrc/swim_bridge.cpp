// swim_bridge.cpp -- the live-node bridge (include/swimbridge.h): one UDP endpoint answering the reference's wire
// protocol for the simulated population.  Host-only: POSIX sockets, the wire codec (swimwire.h) and the PUBLIC entry
// points of swimsim.h -- it holds no pointer into the simulator.  Restates handleUDPMessage.process
// (src/Core.hs:79-117) for datagrams from outside; see the header for the message-by-message rules.
#include "../../include/swimbridge.h"
#include "../../include/swimwire.h"

#include <arpa/inet.h>
#include <netinet/in.h>
#include <poll.h>
#include <sys/socket.h>
#include <unistd.h>

#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

struct swimbridge {
  swimsim_t* sim = nullptr;
  // a sharded cluster behind ONE endpoint (swimbridge_open_cluster): shard k owns members [k * per, (k + 1) * per)
  std::vector<swimsim_t*> shards;
  uint32_t per = 0;
  int fd = -1;
  uint32_t n_members = 0;
  swimbridge_stats_t st{};
  std::string err;
  // IndirectPings forwarded to a node outside the simulation: the Ack with this seqNo from `via` goes to `requester`
  struct Pending { uint32_t seq; sockaddr_in via, requester; };
  std::vector<Pending> pending;
  std::vector<uint8_t> rx, tx;
  std::vector<swimwire_msg_t> in, out;
  bool accept_bare = false;       // swimbridge_accept_bare: datagrams of the literal sender (a bare Message, D11)
};

namespace {

int berr(swimbridge* b, int code, const std::string& m) { if (b) b->err = m; return code; }

// the handle that owns member `id` (the only handle of an unsharded population)
swimsim_t* owner_of(swimbridge* b, uint32_t id) { return b->shards.empty() ? b->sim : b->shards[id / b->per]; }
// a Suspect / Alive / Dead message from outside for `obs` about `id`: to the observer's owner; on a cluster every OTHER shard is
// told too -- the message opens the subject's view row, and a row is a property of the whole cluster (include/swimsim.h)
int inject(swimbridge* b, uint32_t obs, uint32_t id, uint8_t st, uint32_t inc, swimsim_t** failed) {
  swimsim_t* own = owner_of(b, obs);
  *failed = own;
  const int rc = swimsim_inject_rumor(own, obs, id, st, inc);
  if (rc) return rc;
  for (swimsim_t* h : b->shards)
    if (h != own) { const int rc2 = swimsim_note_outside_rumor(h, obs, id); if (rc2 && rc2 != SWIMSIM_ERR_BUFFER) { *failed = h; return rc2; } }
  return SWIMSIM_OK;
}

// "m<id>" -> id (the names the simulator gives its members, include/swimsim.h); false for anything else
bool member_id(const char* name, uint32_t n_members, uint32_t* id) {
  if (name[0] != 'm' || name[1] == 0) return false;
  uint64_t v = 0;
  for (const char* p = name + 1; *p; ++p) {
    if (*p < '0' || *p > '9') return false;
    v = v * 10 + (uint64_t)(*p - '0');
    if (v >= n_members) return false;
  }
  if (name[1] == '0' && name[2] != 0) return false;      // no leading zeros: one name per member
  *id = (uint32_t)v;
  return true;
}

bool same_addr(const sockaddr_in& a, const sockaddr_in& b) { return a.sin_addr.s_addr == b.sin_addr.s_addr && a.sin_port == b.sin_port; }

void name_of(uint32_t id, char* out) { std::snprintf(out, SWIMWIRE_NAME_MAX + 1, "m%u", id); }

// The destination address came out of a datagram (an IndirectPing's target) or is a remote peer's: a send that fails because
// of the ADDRESS (0.0.0.0:0, a broadcast address, an unreachable net) or of a transient local condition (a full socket buffer,
// an interrupted call) is counted and DROPPED -- SEND_DROPPED, so that the caller does not book a datagram that never left --,
// never a failure of the poll (one garbled datagram must not take the bridge down: the reference's receiver dies on bad
// input, D16; this one does not).  Only a broken local socket / host is an error.
constexpr int SEND_DROPPED = 1;
int send_env(swimbridge* b, const std::vector<swimwire_msg_t>& msgs, const sockaddr_in& to) {
  size_t n = 0;
  b->tx.resize(SWIMWIRE_MAX_DATAGRAM);
  const int rc = swimwire_encode(msgs.data(), msgs.size(), b->tx.data(), b->tx.size(), &n);
  if (rc) return berr(b, rc, std::string("encode: ") + swimwire_last_error());
  if (sendto(b->fd, b->tx.data(), n, 0, reinterpret_cast<const sockaddr*>(&to), sizeof to) < 0) {
    const int e = errno;
    if (e == EBADF || e == ENOTSOCK || e == EFAULT || e == ENOMEM || e == ENOBUFS)      // the local socket / host
      return berr(b, SWIMSIM_ERR_DEVICE, std::string("sendto: ") + std::strerror(e));
    b->st.sends_failed++;                                   // EINVAL, EACCES, ENETUNREACH, EHOSTUNREACH, EAGAIN, EINTR, EMSGSIZE, ...
    return SEND_DROPPED;
  }
  b->st.datagrams_out++;
  return SWIMSIM_OK;
}

swimwire_msg_t ack_of(uint32_t seq) { swimwire_msg_t m{}; m.type = SWIMWIRE_ACK; m.seq_no = seq; return m; }

// `Direct (Ack seq []) sender` for simulated member `id` (src/Core.hs:97-99), its piggyback queue riding along (D5)
int answer_ping(swimbridge* b, uint32_t id, uint32_t seq, const sockaddr_in& to) {
  swimsim_member_t mem{};
  const int rc = swimsim_read_member(owner_of(b, id), id, &mem);
  if (rc) return berr(b, rc, std::string("read_member: ") + swimsim_last_error(owner_of(b, id)));
  if (!mem.up) { b->st.pings_unanswered++; return SWIMSIM_OK; }      // a node that is down answers nothing
  b->out.clear();
  b->out.push_back(ack_of(seq));
  for (uint32_t k = 0; k < mem.n_rumors; ++k) {
    const swimsim_rumor_t& r = mem.rumors[k];
    if (r.subject >= b->n_members) continue;      // an entry whose view row was reclaimed names nobody: not on the wire
    swimwire_msg_t m{};
    m.type = r.state == SWIMSIM_SUSPECT ? SWIMWIRE_SUSPECT : r.state == SWIMSIM_DEAD ? SWIMWIRE_DEAD : SWIMWIRE_ALIVE;
    m.incarnation = r.incarnation;
    name_of(r.subject, m.node);
    if (r.state == SWIMSIM_DEAD) name_of(id, m.dead_from);          // the simulator has no addresses: deadFrom = the sender
    if (r.state == SWIMSIM_ALIVE) m.addr = r.subject;               // Alive.addr = the member id
    b->out.push_back(m);
  }
  const int rcs = send_env(b, b->out, to);
  return rcs == SEND_DROPPED ? SWIMSIM_OK : rcs;
}

int handle(swimbridge* b, const sockaddr_in& from, size_t len) {
  size_t n = 0;
  b->in.resize(SWIMWIRE_MAX_MSGS);
  int bare = 0;
  const int drc = b->accept_bare ? swimwire_decode_any(b->rx.data(), len, b->in.data(), b->in.size(), &n, &bare)
                                 : swimwire_decode(b->rx.data(), len, b->in.data(), b->in.size(), &n);
  if (drc != SWIMSIM_OK) { b->st.decode_errors++; return SWIMSIM_OK; }   // D16: dropped
  if (bare) b->st.bare_in++;
  // the simulated member this datagram addresses: the one its Ping names
  uint32_t addressee = 0; bool have_addressee = false;
  for (size_t k = 0; k < n && !have_addressee; ++k)
    if (b->in[k].type == SWIMWIRE_PING) have_addressee = member_id(b->in[k].node, b->n_members, &addressee);
  for (size_t k = 0; k < n; ++k) {
    const swimwire_msg_t& m = b->in[k];
    uint32_t id = 0;
    switch (m.type) {
      case SWIMWIRE_PING:                                           // src/Core.hs:97-101
        if (member_id(m.node, b->n_members, &id)) { b->st.pings++; const int rc = answer_ping(b, id, m.seq_no, from); if (rc) return rc; }
        else b->st.pings_unanswered++;
        break;
      case SWIMWIRE_INDIRECT_PING:                                  // src/Core.hs:105-108 (+ D8, D9)
        b->st.indirect_pings++;
        if (member_id(m.node, b->n_members, &id)) {
          swimsim_member_t mem{};
          const int rc = swimsim_read_member(owner_of(b, id), id, &mem);
          if (rc) return berr(b, rc, std::string("read_member: ") + swimsim_last_error(owner_of(b, id)));
          if (mem.up) { b->out.assign(1, ack_of(m.seq_no)); const int rc2 = send_env(b, b->out, from); if (rc2 < 0) return rc2; if (rc2 == SWIMSIM_OK) b->st.relayed_acks++; }
        } else {
          // target = the HostAddress the reference takes out of a SockAddrInet (src/Core.hs:264-266): already in network
          // byte order as a word (include/swimwire.h) -- reinterpreted, not converted; the port is a PortNumber's value
          sockaddr_in via{}; via.sin_family = AF_INET; via.sin_addr.s_addr = m.target; via.sin_port = htons(m.port);
          swimwire_msg_t p{}; p.type = SWIMWIRE_PING; p.seq_no = m.seq_no; std::memcpy(p.node, m.node, sizeof p.node);
          b->out.assign(1, p);
          const int rc = send_env(b, b->out, via);
          if (rc < 0) return rc;
          if (rc == SEND_DROPPED) break;                                           // nothing went out: nothing to wait for
          if (b->pending.size() >= 4096) b->pending.erase(b->pending.begin());     // the oldest request has timed out long ago
          b->pending.push_back(swimbridge::Pending{m.seq_no, via, from});
        }
        break;
      case SWIMWIRE_ACK: {                                          // src/Core.hs:92-94: the ack rendez-vous; here: the relay of D9
        bool relayed = false;
        for (size_t x = 0; x < b->pending.size(); ++x)
          if (b->pending[x].seq == m.seq_no && same_addr(b->pending[x].via, from)) {
            b->out.assign(1, ack_of(m.seq_no));
            const sockaddr_in to = b->pending[x].requester;
            b->pending.erase(b->pending.begin() + (long)x);
            const int rc = send_env(b, b->out, to);
            if (rc < 0) return rc;
            if (rc == SWIMSIM_OK) b->st.relayed_acks++;
            relayed = true;
            break;
          }
        if (!relayed) b->st.acks_in++;
        break;
      }
      case SWIMWIRE_SUSPECT: case SWIMWIRE_ALIVE: case SWIMWIRE_DEAD:   // src/Core.hs:110-117
        if (member_id(m.node, b->n_members, &id) && m.incarnation >= 0 && m.incarnation <= 0x3FFFFF) {
          const uint32_t obs = have_addressee ? addressee : (id + 1u) % b->n_members;
          const uint8_t st = m.type == SWIMWIRE_SUSPECT ? SWIMSIM_SUSPECT : m.type == SWIMWIRE_DEAD ? SWIMSIM_DEAD : SWIMSIM_ALIVE;
          // more rumours than the simulation takes before its next tick (SWIMSIM_ERR_BUFFER): a flood, dropped and counted
          swimsim_t* failed = nullptr;
          const int rc = inject(b, obs, id, st, (uint32_t)m.incarnation, &failed);
          if (rc == SWIMSIM_ERR_BUFFER) { b->st.rumors_dropped++; break; }
          if (rc) return berr(b, rc, std::string("inject_rumor: ") + swimsim_last_error(failed));
          b->st.rumors_injected++;
        } else b->st.rumors_foreign++;
        break;
      default: break;
    }
  }
  return SWIMSIM_OK;
}

}  // namespace

extern "C" {

static int open_socket(swimbridge* b, const char* bind_ip, uint16_t port, swimbridge_t** out) {
  b->rx.resize(SWIMWIRE_MAX_DATAGRAM + 1);
  b->fd = socket(AF_INET, SOCK_DGRAM, 0);
  sockaddr_in a{}; a.sin_family = AF_INET; a.sin_port = htons(port);
  const bool ip_ok = inet_pton(AF_INET, bind_ip && bind_ip[0] ? bind_ip : "127.0.0.1", &a.sin_addr) == 1;
  int one = 1;
  if (b->fd < 0 || !ip_ok || setsockopt(b->fd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one) < 0 ||   // SO_REUSEADDR: src/Util.hs:59
      bind(b->fd, reinterpret_cast<sockaddr*>(&a), sizeof a) < 0) {
    if (b->fd >= 0) close(b->fd);
    delete b;
    return ip_ok ? SWIMSIM_ERR_DEVICE : SWIMSIM_ERR_INVALID;
  }
  *out = b;
  return SWIMSIM_OK;
}

int swimbridge_open(swimsim_t* sim, const char* bind_ip, uint16_t port, swimbridge_t** out) {
  if (!sim || !out) return SWIMSIM_ERR_INVALID;
  *out = nullptr;
  swimsim_config_t cfg;
  if (swimsim_get_config(sim, &cfg) != SWIMSIM_OK) return SWIMSIM_ERR_INVALID;
  // one endpoint answers for the WHOLE population: a shard owns a slice of it (read_member of another shard's member
  // and inject_rumor are refused there): the shards of a cluster go to swimbridge_open_cluster
  if (cfg.n_shards > 1) return SWIMSIM_ERR_INVALID;
  swimbridge* b = new (std::nothrow) swimbridge();
  if (!b) return SWIMSIM_ERR_NOMEM;
  b->sim = sim; b->n_members = cfg.n_members;
  return open_socket(b, bind_ip, port, out);
}

int swimbridge_open_cluster(swimsim_t* const* shards, uint32_t n_shards, const char* bind_ip, uint16_t port, swimbridge_t** out) {
  if (!shards || !out || n_shards < 2 || n_shards > 16) return SWIMSIM_ERR_INVALID;
  *out = nullptr;
  swimbridge* b = new (std::nothrow) swimbridge();
  if (!b) return SWIMSIM_ERR_NOMEM;
  for (uint32_t k = 0; k < n_shards; ++k) {
    swimsim_config_t cfg;
    // shards[k] must be shard k of n_shards of ONE population with unbounded member maps (messages from outside are not
    // available with view_cap)
    if (!shards[k] || swimsim_get_config(shards[k], &cfg) != SWIMSIM_OK || cfg.n_shards != n_shards || cfg.shard_index != k ||
        cfg.view_cap || (k && cfg.n_members != b->n_members)) { delete b; return SWIMSIM_ERR_INVALID; }
    b->n_members = cfg.n_members;
    b->shards.push_back(shards[k]);
  }
  b->sim = shards[0]; b->per = b->n_members / n_shards;
  return open_socket(b, bind_ip, port, out);
}

int swimbridge_port(const swimbridge_t* b, uint16_t* port) {
  if (!b || !port) return SWIMSIM_ERR_INVALID;
  sockaddr_in a{}; socklen_t l = sizeof a;
  if (getsockname(b->fd, reinterpret_cast<sockaddr*>(&a), &l) < 0) return SWIMSIM_ERR_DEVICE;
  *port = ntohs(a.sin_port);
  return SWIMSIM_OK;
}

int swimbridge_poll(swimbridge_t* b, int timeout_ms, uint32_t max_datagrams) {
  if (!b) return SWIMSIM_ERR_INVALID;
  int handled = 0;
  for (uint32_t k = 0; k < max_datagrams; ++k) {
    pollfd p{b->fd, POLLIN, 0};
    const int pr = poll(&p, 1, k == 0 ? timeout_ms : 0);
    if (pr < 0) { if (errno == EINTR) continue; return berr(b, SWIMSIM_ERR_DEVICE, std::string("poll: ") + std::strerror(errno)); }
    if (pr == 0) break;
    sockaddr_in from{}; socklen_t fl = sizeof from;
    const ssize_t got = recvfrom(b->fd, b->rx.data(), b->rx.size(), 0, reinterpret_cast<sockaddr*>(&from), &fl);
    if (got < 0) { if (errno == EAGAIN || errno == EINTR) continue; return berr(b, SWIMSIM_ERR_DEVICE, std::string("recvfrom: ") + std::strerror(errno)); }
    b->st.datagrams_in++;
    const int rc = handle(b, from, (size_t)got);
    if (rc) return rc;
    handled++;
  }
  return handled;
}

int swimbridge_accept_bare(swimbridge_t* b, int on) {
  if (!b) return SWIMSIM_ERR_INVALID;
  b->accept_bare = on != 0;
  return SWIMSIM_OK;
}

int swimbridge_stats(const swimbridge_t* b, swimbridge_stats_t* out) {
  if (!b || !out) return SWIMSIM_ERR_INVALID;
  *out = b->st;
  return SWIMSIM_OK;
}

const char* swimbridge_last_error(const swimbridge_t* b) { return b ? b->err.c_str() : "no bridge"; }

void swimbridge_close(swimbridge_t* b) {
  if (!b) return;
  if (b->fd >= 0) close(b->fd);
  delete b;
}

}  // extern "C"
