"""Host-side mirror of the reference's types for the hot path (src/Types.hs).

Same names and field meaning as the Haskell records so that the parity tests read
like test/Spec.hs.  Member names are rendered "m<id>" (memberName :: String,
src/Types.hs:62).
"""
from dataclasses import dataclass, field
from enum import IntEnum
from typing import List, Optional, Tuple, Union


class Liveness(IntEnum):
    """data Liveness = IsAliveC | IsSuspectC | IsDeadC   (src/Types.hs:76)"""
    IsAliveC = 0
    IsSuspectC = 1
    IsDeadC = 2


def milliseconds(ms: int) -> int:
    """milliseconds = (1000 *)   (src/Util.hs:23-24) -> Microseconds"""
    return 1000 * ms


@dataclass
class Config:
    """data Config (src/Types.hs:46-51); defaults = parseConfig (src/Util.hs:44-50).

    Only numToGossip and gossipInterval are read by the hot path
    (src/Core.hs:237,239,249,258); the other three are dead fields in the
    reference and are carried, unused, here too."""
    bindHost: str = "udp://127.0.0.1:4002"
    joinHosts: Tuple[str, ...] = ("udp://127.0.0.1:4000",)
    udpBufferSize: int = 65336
    numToGossip: int = 10
    gossipInterval: int = milliseconds(200)


def parseConfig() -> Config:
    """parseConfig :: Either Error Config   (src/Util.hs:44-50) -- a constant."""
    return Config()


@dataclass
class SimConfig:
    """Simulator knobs around the reference Config.

    A value of 0 / None takes the library default documented in include/swimsim.h."""
    cfg: Config = field(default_factory=Config)
    nMembers: int = 128
    seed: int = 1
    probesPerTick: int = 0        # 0 -> numToGossip (D14)
    indirectK: int = 0            # 0 -> numToGossip (D7)
    lossPpm: int = 0
    suspicionTicks: int = 0       # 0 -> 3*ceil(log2 N) (D4)
    retransmitMult: int = 0       # 0 -> 3 (D5)
    maxSubjects: int = 0
    gcTicks: int = 0              # settling horizon (removeDeadNodes); 0 = off, GC_AUTO = S + L + 2
    eventCap: int = 0
    eventMask: int = 0
    inboxCap: int = 0
    device: int = 0
    targetScheme: int = 0         # 0 = kRandomMembers (reference), 1 = robust round-robin (src/Core.hs:232)
    joinPull: int = 0             # 1 = a member that comes up pulls a join host's member map (joinHosts, src/Types.hs:47)
    viewCap: int = 0              # C > 0 = bounded member maps: at most C non-default entries per member, the oldest evicted (include/swimsim.h "Bounded member maps"; `Map String Member`, src/Types.hs:55, with a capacity)
    strictReferenceRules: bool = False   # the LITERAL suspectOrDeadNode' (src/Core.hs:142-187: a Suspect is ignored unless the entry is Alive, a Dead when it is Dead already, at any incarnation) under the canonical order of include/swimsim.h "Strict reference rules" instead of the merge (D13)
    pushPull: bool = False        # with pullTicks: the host of a periodic pull merges the puller's map too (the push half of the commented-out PushPullMsg, src/Types.hs:165,177)
    pullTicks: int = 0            # T > 1 = every up member pulls a random up member's map once per T periods (the commented-out PushPullMsg, src/Types.hs:165,177)


def memberName(member_id: int) -> str:
    return "m%d" % member_id


def memberId(name: str) -> int:
    if not name.startswith("m"):
        raise ValueError("not a simulated member name: %r" % (name,))
    return int(name[1:])


@dataclass(frozen=True)
class Member:
    """data Member (src/Types.hs:62-68) as seen by one observer.  memberLastChange is
    a tick (1 tick = 1 gossipInterval) instead of a UTCTime."""
    memberName: str
    memberAlive: Liveness
    memberIncarnation: int
    memberLastChange: int

    @property
    def id(self) -> int:
        return memberId(self.memberName)


def isAlive(m: Member) -> bool:   # src/Core.hs:33-34
    return m.memberAlive == Liveness.IsAliveC


def isDead(m: Member) -> bool:    # src/Core.hs:36-37
    return m.memberAlive == Liveness.IsDeadC


def notAlive(m: Member) -> bool:  # src/Core.hs:39-40
    return not isAlive(m)


def removeDeadNodes(members: List[Member]) -> List[Member]:
    """removeDeadNodes = Map.filter (not . isDead)   (src/Core.hs:65-67)"""
    return [m for m in members if not isDead(m)]


# --- Message (src/Types.hs:122-145).  The membership messages surface from a tick as events; Ping /
# IndirectPing / Ack are internal to the round and appear only on the wire (swim_amd.wire). ---

@dataclass(frozen=True)
class Ping:
    seqNo: int
    node: str


@dataclass(frozen=True)
class IndirectPing:
    seqNo: int
    target: int
    port: int
    node: str


@dataclass(frozen=True)
class Ack:
    seqNo: int
    payload: List[int] = field(default_factory=list)


@dataclass(frozen=True)
class Suspect:
    incarnation: int
    node: str


@dataclass(frozen=True)
class Alive:
    incarnation: int
    node: str
    addr: int = 0
    port: int = 0


@dataclass(frozen=True)
class Dead:
    incarnation: int
    node: str
    deadFrom: str = ""


Message = Union[Ping, IndirectPing, Ack, Suspect, Alive, Dead]


@dataclass(frozen=True)
class Broadcast:
    """Gossip's `Broadcast Message` (src/Types.hs:42-44): what a node enqueues for
    piggybacking when its view changes (src/Core.hs:119-121,254)."""
    msg: object


@dataclass(frozen=True)
class MembershipEvent:
    """One drained event: at `tick`, `observer` (a member name) changed its entry for the
    subject of `gossip.msg`; `cause` names the reference rule that fired."""
    tick: int
    observer: str
    gossip: Broadcast
    cause: int


def event_message(state: int, incarnation: int, subject: int, observer: int):
    node = memberName(subject)
    if state == Liveness.IsSuspectC:
        return Suspect(incarnation, node)
    if state == Liveness.IsDeadC:
        return Dead(incarnation, node, memberName(observer))
    return Alive(incarnation, node)
