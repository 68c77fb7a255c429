"""Host mirror of the reference's wire layer (src/Types.hs:88-155) over the C codec in libswimsim.so
(include/swimwire.h): `Envelope`, `encode`, `decode` with the reference's names and error style
(`decode :: ByteString -> Either String Envelope`).  Messages are the dataclasses of swim_amd.types.

`datagrams_of` is the byte model of the simulated tick: the compound envelope a member
would put on the wire this period -- its control message followed by its piggyback queue -- so that payload
sizes can be checked against the 255-message / 65 535-byte limits and replayed into a live node."""
import ctypes as C
from typing import List, Optional, Sequence, Tuple

from . import _abi
from .types import Ack, Alive, Dead, IndirectPing, Message, Ping, Suspect, memberName

MAX_MSGS = 255          # numMsgs :: Word8 (src/Types.hs:100)
MAX_DATAGRAM = 65535    # sourceSocket sock 65535 (src/Core.hs:280)


def _lib():
    from . import _lib as loader
    return loader.load()


def _to_c(m: Message) -> _abi.WireMsg:
    w = _abi.WireMsg()
    if isinstance(m, Ping):
        w.type, w.seq_no, w.node = 0, m.seqNo, m.node.encode()
    elif isinstance(m, IndirectPing):
        w.type, w.seq_no, w.target, w.port, w.node = 1, m.seqNo, m.target, m.port, m.node.encode()
    elif isinstance(m, Ack):
        w.type, w.seq_no, w.payload_len = 2, m.seqNo, len(m.payload)
        for k, b in enumerate(m.payload):
            w.payload[k] = b
    elif isinstance(m, Suspect):
        w.type, w.incarnation, w.node = 3, m.incarnation, m.node.encode()
    elif isinstance(m, Alive):
        w.type, w.incarnation, w.node, w.addr, w.port = 4, m.incarnation, m.node.encode(), m.addr, m.port
    elif isinstance(m, Dead):
        w.type, w.incarnation, w.node, w.dead_from = 5, m.incarnation, m.node.encode(), m.deadFrom.encode()
    else:
        raise TypeError("not a Message: %r" % (m,))
    return w


def _from_c(w: _abi.WireMsg) -> Message:
    node = w.node.decode()
    if w.type == 0:
        return Ping(w.seq_no, node)
    if w.type == 1:
        return IndirectPing(w.seq_no, w.target, w.port, node)
    if w.type == 2:
        return Ack(w.seq_no, list(w.payload[: w.payload_len]))
    if w.type == 3:
        return Suspect(w.incarnation, node)
    if w.type == 4:
        return Alive(w.incarnation, node, w.addr, w.port)
    return Dead(w.incarnation, node, w.dead_from.decode())


def encode(msgs: Sequence[Message]) -> bytes:
    """`encode (Envelope msgs)` (src/Types.hs:96-103).  Raises ValueError with the codec's message."""
    lib = _lib()
    arr = (_abi.WireMsg * max(1, len(msgs)))(*[_to_c(m) for m in msgs])
    n = C.c_size_t()
    buf = (C.c_uint8 * MAX_DATAGRAM)()
    rc = lib.wire_encode(arr, len(msgs), buf, MAX_DATAGRAM, C.byref(n))
    if rc != _abi.OK:
        raise ValueError((lib.wire_last_error() or b"").decode())
    return bytes(buf[: n.value])


def decode(data: bytes) -> Tuple[Optional[str], Optional[List[Message]]]:
    """`decode :: ByteString -> Either String Envelope` (src/Types.hs:105-119): (error, None) or (None, messages)."""
    lib = _lib()
    buf = (C.c_uint8 * max(1, len(data))).from_buffer_copy(data) if data else (C.c_uint8 * 1)()
    out = (_abi.WireMsg * MAX_MSGS)()
    n = C.c_size_t()
    rc = lib.wire_decode(buf, len(data), out, MAX_MSGS, C.byref(n))
    if rc != _abi.OK:
        return ((lib.wire_last_error() or b"").decode() or "error %d" % rc, None)
    return (None, [_from_c(out[k]) for k in range(n.value)])


def encode_bare(m: Message) -> bytes:
    """`encode msg` of a bare Message (instance Serialize Message, src/Types.hs:151-155): what the reference's send
    side literally puts on the wire (src/Core.hs:133-134; D11) -- the msgpack body, no type byte."""
    lib = _lib()
    w = _to_c(m)
    n = C.c_size_t()
    buf = (C.c_uint8 * MAX_DATAGRAM)()
    rc = lib.wire_encode_bare(C.byref(w), buf, MAX_DATAGRAM, C.byref(n))
    if rc != _abi.OK:
        raise ValueError((lib.wire_last_error() or b"").decode())
    return bytes(buf[: n.value])


def decode_any(data: bytes) -> Tuple[Optional[str], Optional[List[Message]], bool]:
    """An Envelope or a bare Message (the literal sender's datagram): (error, None, bare) or (None, messages, bare)."""
    lib = _lib()
    buf = (C.c_uint8 * max(1, len(data))).from_buffer_copy(data) if data else (C.c_uint8 * 1)()
    out = (_abi.WireMsg * MAX_MSGS)()
    n = C.c_size_t()
    bare = C.c_int()
    rc = lib.wire_decode_any(buf, len(data), out, MAX_MSGS, C.byref(n), C.byref(bare))
    if rc != _abi.OK:
        return ((lib.wire_last_error() or b"").decode() or "error %d" % rc, None, bool(bare.value))
    return (None, [_from_c(out[k]) for k in range(n.value)], bool(bare.value))


def rumor_message(subject: int, incarnation: int, state: int, sender: int) -> Message:
    """A piggybacked rumour as the reference's Message (state: 0 Alive, 1 Suspect, 2 Dead).  The simulator has
    no addresses: Alive.addr is the member id, the port the reference's 4000; Dead.deadFrom names the sender."""
    if state == 1:
        return Suspect(incarnation, memberName(subject))
    if state == 2:
        return Dead(incarnation, memberName(subject), memberName(sender))
    return Alive(incarnation, memberName(subject), subject, 4000)


def datagram_of(sim, src: int, control: Message) -> bytes:
    """The datagram member `src` sends this period for `control` (a Ping / Ack / IndirectPing): the control
    message and src's current piggyback queue as one compound envelope (D5; src/Types.hs:96-103)."""
    m = sim.readMember(src)
    msgs = [control] + [rumor_message(subj, inc, st, src) for (subj, inc, st, _tx) in m["rumors"]]
    return encode(msgs)
