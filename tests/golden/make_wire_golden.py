"""Golden datagrams of the wire codec (include/swimwire.h), derived BY HAND from the msgpack specification and the framing
of src/Types.hs:96-119 -- not produced by the codec under test: every byte below is written out with the rule it comes
from.  The reference holds no golden bytes (test/Spec.hs:77-96 are round trips), so these pin the codec's side of the
interoperability contract: bodies are msgpack maps {"tag": constructor, record fields...} as `packAeson . toJSON` gives them
(aeson's default sum encoding for records: TaggedObject with tagFieldName "tag"); keys in declaration order here (the
reference's own order is a HashMap's and unknown without GHC: `permuted` holds one datagram in another order, which the
decoder must accept).  Values: the ones of test/Spec.hs:77-96 for Ping / IndirectPing / Ack, one of each width class else.

    python tests/golden/make_wire_golden.py      # rewrites tests/golden/wire_*.bin and wire_golden.json
"""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))


def fixstr(s):                       # msgpack: 101xxxxx + bytes, for strings shorter than 32 bytes
    b = s.encode()
    assert len(b) < 32
    return bytes([0xA0 | len(b)]) + b


def fixmap(n):                       # msgpack: 1000xxxx, maps of up to 15 pairs
    return bytes([0x80 | n])


def fixarray(n):                     # msgpack: 1001xxxx
    return bytes([0x90 | n])


def posfixint(v):                    # msgpack: 0xxxxxxx, 0..127
    assert 0 <= v < 128
    return bytes([v])


def uint8(v): return b"\xcc" + bytes([v])                       # 128..255
def uint16(v): return b"\xcd" + v.to_bytes(2, "big")            # 256..65535
def uint32(v): return b"\xce" + v.to_bytes(4, "big")            # 65536..2^32-1
def negfixint(v): return bytes([v & 0xFF])                      # 111xxxxx, -32..-1


BODY = {
    # Ping { seqNo = 1, node = "a" }                                                    (test/Spec.hs:80)
    "ping": fixmap(3) + fixstr("tag") + fixstr("Ping") + fixstr("seqNo") + posfixint(1) + fixstr("node") + fixstr("a"),
    # IndirectPing { seqNo = 2, target = 1, port = 4000, node = "b" }                   (test/Spec.hs:81): 4000 needs uint16
    "indirect_ping": fixmap(5) + fixstr("tag") + fixstr("IndirectPing") + fixstr("seqNo") + posfixint(2) + fixstr("target") + posfixint(1)
                     + fixstr("port") + uint16(4000) + fixstr("node") + fixstr("b"),
    # Ack { seqNo = 2, payload = [] }                                                    (test/Spec.hs:82)
    "ack": fixmap(3) + fixstr("tag") + fixstr("Ack") + fixstr("seqNo") + posfixint(2) + fixstr("payload") + fixarray(0),
    # Suspect { incarnation = 300, node = "m17" }: 300 needs uint16
    "suspect": fixmap(3) + fixstr("tag") + fixstr("Suspect") + fixstr("incarnation") + uint16(300) + fixstr("node") + fixstr("m17"),
    # Alive { incarnation = 70000, node = "m1048575", addr = 16777343 (127.0.0.1 as the HostAddress word on x86), port = 4001 }
    "alive": fixmap(5) + fixstr("tag") + fixstr("Alive") + fixstr("incarnation") + uint32(70000) + fixstr("node") + fixstr("m1048575")
             + fixstr("addr") + uint32(16777343) + fixstr("port") + uint16(4001),
    # Dead { incarnation = 200, node = "m9", deadFrom = "m3" }: 200 needs uint8
    "dead": fixmap(4) + fixstr("tag") + fixstr("Dead") + fixstr("incarnation") + uint8(200) + fixstr("node") + fixstr("m9")
            + fixstr("deadFrom") + fixstr("m3"),
    # Ack { seqNo = 4294967295, payload = [0, 127, 128, 255] }: the widest seqNo, payload bytes on both sides of 128
    "ack_payload": fixmap(3) + fixstr("tag") + fixstr("Ack") + fixstr("seqNo") + uint32(4294967295) + fixstr("payload") + fixarray(4)
                   + posfixint(0) + posfixint(127) + uint8(128) + uint8(255),
    # Suspect { incarnation = -1, node = "x" }: a Haskell Int may be negative on the wire (the simulator refuses it later)
    "suspect_negative": fixmap(3) + fixstr("tag") + fixstr("Suspect") + fixstr("incarnation") + negfixint(-1) + fixstr("node") + fixstr("x"),
}
MSGS = {
    "ping": {"type": "Ping", "seqNo": 1, "node": "a"},
    "indirect_ping": {"type": "IndirectPing", "seqNo": 2, "target": 1, "port": 4000, "node": "b"},
    "ack": {"type": "Ack", "seqNo": 2, "payload": []},
    "suspect": {"type": "Suspect", "incarnation": 300, "node": "m17"},
    "alive": {"type": "Alive", "incarnation": 70000, "node": "m1048575", "addr": 16777343, "port": 4001},
    "dead": {"type": "Dead", "incarnation": 200, "node": "m9", "deadFrom": "m3"},
    "ack_payload": {"type": "Ack", "seqNo": 4294967295, "payload": [0, 127, 128, 255]},
    "suspect_negative": {"type": "Suspect", "incarnation": -1, "node": "x"},
}
MSG_INDEX = {"Ping": 0, "IndirectPing": 1, "Ack": 2, "Suspect": 3, "Alive": 4, "Dead": 5}   # msgIndex (src/Types.hs:169-178)
COMPOUND = 6                                                                                # fromEnum CompoundMsg


def single(name):                    # put (Envelope (msg :| [])) = putWord8 (msgIndex msg) >> put msg       (src/Types.hs:97)
    return bytes([MSG_INDEX[MSGS[name]["type"]]]) + BODY[name]


def compound(names):                 # [CompoundMsg][n u8][len u16be x n][bodies]                            (src/Types.hs:98-103)
    out = bytes([COMPOUND, len(names)])
    for n in names:
        out += len(BODY[n]).to_bytes(2, "big")
    for n in names:
        out += BODY[n]
    return out


def main():
    files = {}
    for name in BODY:
        files["wire_%s.bin" % name] = {"bytes": single(name), "msgs": [name], "form": "envelope"}
        files["wire_bare_%s.bin" % name] = {"bytes": BODY[name], "msgs": [name], "form": "bare"}     # `encode msg`, src/Core.hs:133-134 (D11)
    # what a simulated member puts on the wire in a period: its control message + its piggyback queue (row a18)
    files["wire_compound.bin"] = {"bytes": compound(["ping", "suspect", "alive", "dead"]), "msgs": ["ping", "suspect", "alive", "dead"], "form": "envelope"}
    # test/Spec.hs:83-85,96: [ping, ack, ping2, ack2] with ping2 = Ping 3 "b", ack2 = Ack 4 []
    BODY["ping2"] = fixmap(3) + fixstr("tag") + fixstr("Ping") + fixstr("seqNo") + posfixint(3) + fixstr("node") + fixstr("b")
    BODY["ack2"] = fixmap(3) + fixstr("tag") + fixstr("Ack") + fixstr("seqNo") + posfixint(4) + fixstr("payload") + fixarray(0)
    MSGS["ping2"] = {"type": "Ping", "seqNo": 3, "node": "b"}
    MSGS["ack2"] = {"type": "Ack", "seqNo": 4, "payload": []}
    files["wire_spec_hs_compound.bin"] = {"bytes": compound(["ping", "ack", "ping2", "ack2"]), "msgs": ["ping", "ack", "ping2", "ack2"], "form": "envelope"}
    # another key order (a HashMap's, say) and wider integers than needed: decodes to the same message, is NOT what the encoder writes
    permuted = fixmap(5) + fixstr("node") + fixstr("b") + fixstr("port") + uint32(4000) + fixstr("seqNo") + uint8(2) + fixstr("tag") + fixstr("IndirectPing") \
        + fixstr("target") + uint16(1)
    files["wire_permuted_indirect_ping.bin"] = {"bytes": bytes([1]) + permuted, "msgs": ["indirect_ping"], "form": "envelope", "decode_only": True}
    manifest = {"messages": MSGS, "files": {}}
    for fn, f in sorted(files.items()):
        with open(os.path.join(HERE, fn), "wb") as fh:
            fh.write(f["bytes"])
        manifest["files"][fn] = {k: v for k, v in f.items() if k != "bytes"}
        manifest["files"][fn]["hex"] = f["bytes"].hex()
    with open(os.path.join(HERE, "wire_golden.json"), "w") as fh:
        json.dump(manifest, fh, indent=1, sort_keys=True)
    print("wrote %d datagrams" % len(files))


if __name__ == "__main__":
    main()
