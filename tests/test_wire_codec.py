"""The wire codec (include/swimwire.h) -- host-only code of libswimsim.so, so
these tests need no GPU.  test/Spec.hs:77-96 restated (round trips of Ping / IndirectPing alone and of a
4-message compound), the framing bytes of src/Types.hs:96-119 checked literally, the bodies cross-checked
with an independent msgpack implementation (the `msgpack` Python package), the reference's decode failures,
and the 255-message / 65 535-byte bounds of a piggybacked datagram."""
import struct

import msgpack
import pytest

import __graft_entry__ as g
from swim_amd import Ack, Alive, Dead, IndirectPing, Ping, Suspect
from swim_amd import wire


@pytest.fixture(scope="module", autouse=True)
def built():
    g.build()


def back_and_forth(msgs):
    """backAndForthForeverAndEver msgs = decode (encode $ envelope msgs)   (test/Spec.hs:64-65)"""
    return wire.decode(wire.encode(msgs))


def test_spec_hs_wire_protocol_encodes_and_decodes():
    """test/Spec.hs:77-96, same values."""
    ping = Ping(seqNo=1, node="a")
    indirectPing = IndirectPing(seqNo=2, target=1, port=4000, node="b")
    ack = Ack(seqNo=2, payload=[])
    ping2 = Ping(seqNo=3, node="b")
    ack2 = Ack(seqNo=4, payload=[])
    msgs = [ping, ack, ping2, ack2]
    assert back_and_forth([ping]) == (None, [ping])
    assert back_and_forth([indirectPing]) == (None, [indirectPing])
    assert back_and_forth(msgs) == (None, msgs)


def test_framing_bytes_are_the_reference_layout():
    """single: [msgIndex][body]; compound: [6][n][u16be lengths][bodies]   (src/Types.hs:96-103)"""
    ping = Ping(1, "a")
    one = wire.encode([ping])
    assert one[0] == 0
    body = one[1:]
    assert msgpack.unpackb(body, raw=False) == {"tag": "Ping", "seqNo": 1, "node": "a"}
    msgs = [ping, Ack(2, []), Suspect(7, "m12"), Alive(3, "m5", 5, 4000), Dead(9, "m77", "m1"),
            IndirectPing(2, 1, 4000, "b")]
    data = wire.encode(msgs)
    assert data[0] == 6 and data[1] == len(msgs)
    lens = struct.unpack(">%dH" % len(msgs), data[2:2 + 2 * len(msgs)])
    off = 2 + 2 * len(msgs)
    expect = [{"tag": "Ping", "seqNo": 1, "node": "a"}, {"tag": "Ack", "seqNo": 2, "payload": []},
              {"tag": "Suspect", "incarnation": 7, "node": "m12"},
              {"tag": "Alive", "incarnation": 3, "node": "m5", "addr": 5, "port": 4000},
              {"tag": "Dead", "incarnation": 9, "node": "m77", "deadFrom": "m1"},
              {"tag": "IndirectPing", "seqNo": 2, "target": 1, "port": 4000, "node": "b"}]
    for ln, want in zip(lens, expect):
        assert msgpack.unpackb(data[off:off + ln], raw=False) == want
        off += ln
    assert off == len(data)
    assert wire.encode([msgs[2]])[0] == 3 and wire.encode([msgs[4]])[0] == 5       # msgIndex


def test_decoder_accepts_what_another_msgpack_writer_produces():
    """The reference's key order (a HashMap's) and integer widths are not reproducible here: any order, any
    width, unknown extra fields must parse (packAeson / unpackAeson contract)."""
    def body(d):
        return msgpack.packb(d, use_bin_type=False)
    for d, want in (
        ({"node": "zeta", "seqNo": 70000, "tag": "Ping"}, Ping(70000, "zeta")),
        ({"payload": [1, 2, 255], "tag": "Ack", "seqNo": 0, "extra": {"x": [1, None, True]}}, Ack(0, [1, 2, 255])),
        ({"port": 1, "addr": 0xFFFFFFFF, "node": "m1", "incarnation": -5, "tag": "Alive"}, Alive(-5, "m1", 0xFFFFFFFF, 1)),
        ({"deadFrom": "", "tag": "Dead", "node": "x" * 40, "incarnation": 2 ** 40}, Dead(2 ** 40, "x" * 40, "")),
    ):
        assert wire.decode(bytes([0]) + body(d)) == (None, [want])
    # widest integer encodings for small values
    b = b"\x83\xa3tag\xa4Ping\xa5seqNo\xcf" + struct.pack(">Q", 9) + b"\xa4node\xa1a"
    assert wire.decode(b"\x00" + b) == (None, [Ping(9, "a")])
    bodies = [body({"tag": "Suspect", "node": "m3", "incarnation": 1}), body({"seqNo": 5, "tag": "Ack", "payload": []})]
    comp = bytes([6, 2]) + struct.pack(">HH", len(bodies[0]), len(bodies[1])) + bodies[0] + bodies[1]
    assert wire.decode(comp) == (None, [Suspect(1, "m3"), Ack(5, [])])


def test_decode_failures_of_the_reference():
    """src/Types.hs:105-119: truncated compound, zero messages, unknown type, unparsable body."""
    assert wire.decode(b"")[0]
    assert "invalid message type" in wire.decode(b"\x09\x80")[0]
    assert wire.decode(bytes([6, 3, 0, 1]))[0] == "compound message is truncated"
    assert wire.decode(bytes([6, 0]))[0] == "compound mesage with zero messages"
    assert "Could not parse" in wire.decode(b"\x00\x01\x02")[0]
    assert "missing" in wire.decode(b"\x00" + msgpack.packb({"tag": "Ping", "seqNo": 1}))[0]
    assert "out of bounds" in wire.decode(b"\x00" + msgpack.packb({"tag": "Ping", "seqNo": 2 ** 32, "node": "a"}))[0]
    assert "unknown constructor" in wire.decode(b"\x00" + msgpack.packb({"tag": "PushPull"}))[0]
    good = wire.encode([Ping(1, "a"), Ack(1, [])])
    assert wire.decode(good[:-1])[0]                                   # isolate: body shorter than its length
    body = msgpack.packb({"tag": "Ack", "seqNo": 1, "payload": []}) + b"\x00"
    assert "trailing" in wire.decode(bytes([6, 1]) + struct.pack(">H", len(body)) + body)[0]


def test_envelope_bounds_row_a18():
    """<= 255 messages (numMsgs :: Word8), <= 65 535 bytes (src/Core.hs:280); an Envelope is NonEmpty."""
    with pytest.raises(ValueError, match="at least one"):
        wire.encode([])
    many = [Suspect(k, "m%d" % k) for k in range(255)]
    err, got = back_and_forth(many)
    assert err is None and got == many
    with pytest.raises(ValueError, match="255"):
        wire.encode(many + [Suspect(1, "x")])
    fat = [Ack(k, list(range(255))) for k in range(200)]               # ~410 bytes each
    with pytest.raises(ValueError, match="65 535"):
        wire.encode(fat)
    # the simulator's piggybacked datagram: control message + 8 rumours at the widest values it can hold
    # (27-bit member ids, 22-bit incarnations) stays two orders of magnitude below the limit
    worst = [IndirectPing(2 ** 32 - 1, 2 ** 27 - 1, 65535, "m%d" % (2 ** 27 - 1))]
    worst += [Dead(2 ** 22 - 1, "m%d" % (2 ** 27 - 1 - k), "m%d" % (2 ** 27 - 1)) for k in range(8)]
    assert len(wire.encode(worst)) < 700


def test_datagram_of_a_simulated_member(oracle_abi):
    """Byte model of a tick: what a member puts on the wire = its control message + its piggyback queue
    (the oracle handle stands in for the device handle: the codec itself is host code)."""
    from swim_amd import Config, Sim, SimConfig
    s = Sim.create(oracle_abi, SimConfig(cfg=Config(numToGossip=3), nMembers=64, seed=2, suspicionTicks=5))
    s.crash(7, 1); s.crash(9, 1)
    s.step(6)
    m = s.readMember(0)
    assert m["rumors"]
    data = wire.datagram_of(s, 0, Ping(int(s.tick) + 1, "m3"))
    err, msgs = wire.decode(data)
    assert err is None and msgs[0] == Ping(int(s.tick) + 1, "m3") and len(msgs) == 1 + len(m["rumors"])
    assert {(type(x).__name__, x.node) for x in msgs[1:]} <= {("Suspect", "m7"), ("Suspect", "m9"), ("Dead", "m7"), ("Dead", "m9")}
    assert len(data) <= wire.MAX_DATAGRAM and len(msgs) <= wire.MAX_MSGS


# ---- golden datagrams: bytes written out by hand from the msgpack specification (tests/golden/make_wire_golden.py) -------------
def _golden():
    import json
    import os
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    man = json.load(open(os.path.join(here, "wire_golden.json")))
    ctor = {"Ping": Ping, "IndirectPing": IndirectPing, "Ack": Ack, "Suspect": Suspect, "Alive": Alive, "Dead": Dead}
    msgs = {k: ctor[v["type"]](**{f: x for f, x in v.items() if f != "type"}) for k, v in man["messages"].items()}
    for fn, f in sorted(man["files"].items()):
        data = open(os.path.join(here, fn), "rb").read()
        assert data.hex() == f["hex"], "%s differs from its manifest (rerun tests/golden/make_wire_golden.py)" % fn
        yield fn, data, [msgs[m] for m in f["msgs"]], f["form"], f.get("decode_only", False)


def test_golden_datagrams_decode_and_encode_byte_for_byte():
    """All six constructors, single and compound framing, the bare form of the literal sender (D11): the codec decodes
    the hand-derived bytes to the messages they spell and -- except for the datagram in a foreign key order -- writes
    exactly those bytes."""
    seen = set()
    for fn, data, msgs, form, decode_only in _golden():
        if form == "envelope":
            assert wire.decode(data) == (None, msgs), fn
            assert wire.decode_any(data) == (None, msgs, False), fn
            if not decode_only:
                assert wire.encode(msgs) == data, fn
        else:
            assert wire.decode_any(data) == (None, msgs, True), fn
            assert wire.encode_bare(msgs[0]) == data, fn
            err, _ = wire.decode(data)                     # `decode` proper wants an Envelope: the reference's two sides
            assert err is not None and "invalid message type" in err, fn     # cannot talk to each other (D11)
        seen.update(type(m).__name__ for m in msgs)
    assert seen == {"Ping", "IndirectPing", "Ack", "Suspect", "Alive", "Dead"}


def test_golden_generator_is_reproducible(tmp_path):
    """The committed .bin files are what the committed script writes."""
    import importlib.util
    import os
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("make_wire_golden", os.path.join(here, "make_wire_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.HERE = str(tmp_path)
    mod.main()
    for fn in os.listdir(str(tmp_path)):
        assert open(os.path.join(str(tmp_path), fn), "rb").read() == open(os.path.join(here, fn), "rb").read(), fn
