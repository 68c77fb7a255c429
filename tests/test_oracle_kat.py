"""Known-answer tests that PIN THE ORACLE against the reference's own test suite
(test/Spec.hs of jpfuentes2/swim) and against the reading of src/Core.hs.  CPU only.

Fixture (test/Spec.hs:45-56, 31-34): three members "alive"/"suspect"/"dead" at incarnation 0 and
self = "myself".  Here: ids 0=alive, 1=suspect, 2=dead, 3=myself."""
import ctypes as C

import pytest

from swim_amd import Config, Liveness, Member, Sim, SimConfig, removeDeadNodes
from tests import oracle_binding as ob

ALIVE, SUSPECT, DEAD, MYSELF = 0, 1, 2, 3


def with_store(oracle_abi, n=4, seed=1, **kw):
    return Sim.create(oracle_abi, SimConfig(cfg=Config(numToGossip=3), nMembers=n, seed=seed, eventMask=0x1F, **kw))


def default_members(s):
    s.setView(MYSELF, SUSPECT, Liveness.IsSuspectC, 0)
    s.setView(MYSELF, DEAD, Liveness.IsDeadC, 0)


# ---- Core.removeDeadNodes (test/Spec.hs:98-106) ---------------------------------------------
def test_remove_dead_nodes(oracle_abi):
    s = with_store(oracle_abi)
    default_members(s)
    view = s.members(MYSELF)
    assert [m.memberName for m in view] == ["m1", "m2"]         # non-default entries only
    full = [Member("m0", Liveness.IsAliveC, 0, 0)] + view       # {alive, suspect, dead}
    kept = removeDeadNodes(full)
    assert "m2" not in [m.memberName for m in kept] and len(kept) == 2
    # the oracle's own C restatement of Map.filter (not . isDead)
    from swim_amd import _abi
    arr = (_abi.ViewEntry * 3)()
    for k, st in enumerate((0, 1, 2)):
        arr[k].subject = k; arr[k].state = st
    n = oracle_abi.lib.swimoracle_remove_dead_nodes(arr, 3)
    assert n == 2 and [arr[k].subject for k in range(n)] == [0, 1]


# ---- Core.kRandomMembers (test/Spec.hs:108-139) ----------------------------------------------
def test_k_random_members_takes_no_nodes_if_n_is_0(oracle_abi):
    s = with_store(oracle_abi); default_members(s)
    assert s.kRandomMembers(MYSELF, 0, [ALIVE, SUSPECT, DEAD]) == []


def test_k_random_members_filters_non_alive_nodes(oracle_abi):
    s = with_store(oracle_abi); default_members(s)
    rand = s.kRandomMembers(MYSELF, 3, [])
    assert len(rand) == 1 and rand[0] == ALIVE                  # `head rand shouldBe head ms`


def test_k_random_members_filters_exclusion_nodes(oracle_abi):
    s = with_store(oracle_abi); default_members(s)
    assert s.kRandomMembers(MYSELF, 3, [ALIVE]) == []


def test_k_random_members_shuffles(oracle_abi):
    total, n = 200, 50
    s = with_store(oracle_abi, n=total + 1)                     # 200 alive + self
    rand = s.kRandomMembers(total, n, [])
    assert len(rand) == n and len(set(rand)) == n
    assert rand != list(range(n))                               # `rand shouldNotBe alives`
    assert all(0 <= r < total for r in rand)


def test_k_random_members_uniform(oracle_abi):
    """Not in the reference: the draw is uniform over the eligible members (chi-square)."""
    n = 64
    counts = [0] * n
    for seed in range(400):
        s = with_store(oracle_abi, n=n + 1, seed=seed)
        for r in s.kRandomMembers(n, 8, []):
            counts[r] += 1
        s.close()
    expect = 400 * 8 / n
    chi2 = sum((c - expect) ** 2 / expect for c in counts)
    assert chi2 < 120, chi2                                     # 63 dof: p(>120) ~ 2e-5


# ---- Core.handleUDPMessage (test/Spec.hs:141-183) --------------------------------------------
def _msg(**kw):
    m = ob.OMsg()
    for k, v in kw.items():
        setattr(m, k, v)
    return m


def test_gets_ping_for_us_responds_with_ack(oracle_abi):
    s = with_store(oracle_abi)
    sender = ALIVE
    out = ob.process(s, MYSELF, sender, _msg(type=ob.MSG_PING, seq_no=1, node=MYSELF))
    assert len(out) == 1
    assert (out[0].type, out[0].seq_no, out[0].to, out[0].broadcast) == (ob.MSG_ACK, 1, sender, 0)


def test_gets_ping_for_someone_else_ignores(oracle_abi):
    s = with_store(oracle_abi)
    assert ob.process(s, MYSELF, ALIVE, _msg(type=ob.MSG_PING, seq_no=1, node=DEAD)) == []


def test_gets_ack_invokes_handler_no_gossip(oracle_abi):
    """`pending` in the reference (test/Spec.hs:160-164); the rule is src/Core.hs:92-94: []"""
    s = with_store(oracle_abi)
    assert ob.process(s, MYSELF, ALIVE, _msg(type=ob.MSG_ACK, seq_no=1)) == []


def test_gets_indirect_ping_sends_ping_literal_d8(oracle_abi):
    """test/Spec.hs:166-174, LITERAL behaviour: storeIncarnation + 1, Ping seqNo = new
    incarnation, node = requested node, sent to the target."""
    s = with_store(oracle_abi)
    before = s.readMember(MYSELF)["incarnation"]
    out = ob.process(s, MYSELF, ALIVE, _msg(type=ob.MSG_INDIRECT_PING, seq_no=1, target=SUSPECT, node=SUSPECT),
                     literal_d8=True)
    after = s.readMember(MYSELF)["incarnation"]
    assert before + 1 == after
    assert len(out) == 1
    assert (out[0].type, out[0].seq_no, out[0].node, out[0].to) == (ob.MSG_PING, 1, SUSPECT, SUSPECT)


def test_gets_indirect_ping_sends_ping_tick_semantics(oracle_abi):
    """Deliberate divergence D8: the tick relays the REQUESTER's seqNo and does not touch the
    incarnation counter."""
    s = with_store(oracle_abi)
    out = ob.process(s, MYSELF, ALIVE, _msg(type=ob.MSG_INDIRECT_PING, seq_no=7, target=SUSPECT, node=SUSPECT))
    assert s.readMember(MYSELF)["incarnation"] == 0
    assert (out[0].type, out[0].seq_no, out[0].node, out[0].to) == (ob.MSG_PING, 7, SUSPECT, SUSPECT)


# the three `pending` examples (test/Spec.hs:176-183): rules read off src/Core.hs:142-218
def test_gets_suspect(oracle_abi):
    s = with_store(oracle_abi)
    out = ob.process(s, MYSELF, ALIVE, _msg(type=ob.MSG_SUSPECT, incarnation=0, node=ALIVE))
    assert [(m.memberName, m.memberAlive, m.memberIncarnation) for m in s.members(MYSELF)] == [("m0", Liveness.IsSuspectC, 0)]
    assert len(out) == 1 and out[0].broadcast == 1 and out[0].type == ob.MSG_SUSPECT   # re-broadcast (:179)
    # same message again: already suspect -> ignored (livenessCheck, :183)
    assert ob.process(s, MYSELF, ALIVE, _msg(type=ob.MSG_SUSPECT, incarnation=0, node=ALIVE)) == []


def test_gets_dead(oracle_abi):
    s = with_store(oracle_abi)
    default_members(s)
    out = ob.process(s, MYSELF, ALIVE, _msg(type=ob.MSG_DEAD, incarnation=0, node=SUSPECT, dead_from=ALIVE))
    assert s.members(MYSELF)[0].memberAlive == Liveness.IsDeadC
    assert len(out) == 1 and out[0].broadcast == 1
    assert ob.process(s, MYSELF, ALIVE, _msg(type=ob.MSG_DEAD, incarnation=0, node=DEAD)) == []   # already dead (:184)


def test_gets_suspect_about_self_refutes(oracle_abi):
    """src/Core.hs:155-166 with D10's fix: incarnation := rumour's + 1, answer Alive."""
    s = with_store(oracle_abi)
    out = ob.process(s, MYSELF, ALIVE, _msg(type=ob.MSG_SUSPECT, incarnation=0, node=MYSELF))
    assert s.readMember(MYSELF)["incarnation"] == 1
    assert len(out) == 1 and (out[0].type, out[0].incarnation, out[0].node, out[0].broadcast) == (ob.MSG_ALIVE, 1, MYSELF, 1)
    assert s.members(MYSELF) == []                               # never marks itself
    # a stale rumour (incarnation below ours) is ignored (:151)
    assert ob.process(s, MYSELF, ALIVE, _msg(type=ob.MSG_DEAD, incarnation=0, node=MYSELF)) == []


def test_gets_alive(oracle_abi):
    """aliveNode is unwritten in the reference (throws, D6); the rule implemented is the SWIM /
    memberlist one: Alive@i overrides Suspect@j and Alive@j iff i > j."""
    s = with_store(oracle_abi)
    default_members(s)
    assert ob.process(s, MYSELF, ALIVE, _msg(type=ob.MSG_ALIVE, incarnation=0, node=SUSPECT)) == []
    out = ob.process(s, MYSELF, ALIVE, _msg(type=ob.MSG_ALIVE, incarnation=1, node=SUSPECT))
    assert len(out) == 1 and out[0].broadcast == 1
    v = s.members(MYSELF)
    assert (v[0].memberName, v[0].memberAlive, v[0].memberIncarnation) == ("m1", Liveness.IsAliveC, 1)


# ---- the state rule vs the literal reference rule ------------------------------------------------
def test_merge_rule_vs_literal_reference_rule(oracle_abi):
    """Exhaustive comparison on incarnations 0..3: the commutative merge equals the literal
    suspectOrDeadNode' (src/Core.hs:151-152,182-184) except in the documented D13 cases:
    the literal rule drops a Suspect/Dead at a HIGHER incarnation when the entry is already
    Suspect/Dead (Suspect) or Dead (Dead), and ties on (inc, state) order."""
    lib = oracle_abi.lib
    diffs = []
    for cinc in range(4):
        for cst in (0, 1, 2):
            for minc in range(4):
                for mst in (1, 2):
                    cur, msg = (cinc << 2) | cst, (minc << 2) | mst
                    lit, mer = lib.swimoracle_reference_rule(cur, msg), lib.swimoracle_merge_rule(cur, msg)
                    if lit != mer:
                        diffs.append((cinc, cst, minc, mst, lit, mer))
    for (cinc, cst, minc, mst, lit, mer) in diffs:
        # every difference: message at a strictly higher incarnation (or Dead@same over Dead) that the
        # literal code ignores because the entry is not Alive (Suspect) / already Dead (Dead)
        if mst == 1:
            assert cst != 0 and minc > cinc, (cinc, cst, minc, mst)
        else:
            assert cst == 2 and minc > cinc, (cinc, cst, minc, mst)
        assert lit == (cinc << 2) | cst and mer == (minc << 2) | mst
    # and wherever the entry is Alive the two rules agree completely
    assert not [d for d in diffs if d[1] == 0]


def test_merge_rule_is_commutative_and_idempotent(oracle_abi):
    lib = oracle_abi.lib
    keys = [(i << 2) | st for i in range(3) for st in (0, 1, 2)]
    for a in keys:
        for b in keys:
            for c in keys:
                x = lib.swimoracle_merge_rule(lib.swimoracle_merge_rule(a, b), c)
                y = lib.swimoracle_merge_rule(lib.swimoracle_merge_rule(a, c), b)
                assert x == y
            assert lib.swimoracle_merge_rule(lib.swimoracle_merge_rule(a, b), b) == lib.swimoracle_merge_rule(a, b)


# ---- configure / parseConfig (src/Util.hs:44-50,103-107) -----------------------------------------
def test_default_config_is_parse_config(oracle_abi):
    from swim_amd import _abi, parseConfig
    c = _abi.Config()
    assert oracle_abi.default_config(C.byref(c)) == 0
    ref = parseConfig()
    assert (c.num_to_gossip, c.gossip_interval_us) == (ref.numToGossip, ref.gossipInterval) == (10, 200000)
    assert ref.bindHost == "udp://127.0.0.1:4002" and ref.joinHosts == ("udp://127.0.0.1:4000",) and ref.udpBufferSize == 65336


def test_configure_returns_left_on_bad_config(oracle_abi):
    err, sim = Sim.configure(oracle_abi, SimConfig(nMembers=1))
    assert sim is None and "n_members" in err
