"""Sim: the host-side handle over the C ABI (include/swimsim.h).

`configure` mirrors the reference's `configure :: IO (Either Error Store)`
(src/Util.hs:103-107): it returns (error_string, None) or (None, Sim) instead of
raising.  Runtime failures raise SwimError, the counterpart of the reference's
`fail`/ioError style (src/Core.hs:87).
"""
import ctypes as C
from typing import Iterable, List, Optional, Sequence, Tuple

from . import _abi
from .types import (Broadcast, Liveness, Member, MembershipEvent, SimConfig, event_message,
                    memberName)


class SwimError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("swimsim status %d: %s" % (status, message))
        self.status = status
        self.message = message


def _to_c_config(sc: SimConfig, shard_index: int = 0, n_shards: int = 1) -> _abi.Config:
    c = _abi.Config()
    c.struct_size = C.sizeof(_abi.Config)
    c.abi_version = _abi.ABI_VERSION
    c.num_to_gossip = sc.cfg.numToGossip
    c.gossip_interval_us = sc.cfg.gossipInterval
    c.n_members = sc.nMembers
    c.seed = sc.seed
    c.probes_per_tick = sc.probesPerTick
    c.indirect_k = sc.indirectK
    c.loss_ppm = sc.lossPpm
    c.suspicion_ticks = sc.suspicionTicks
    c.retransmit_mult = sc.retransmitMult
    c.max_subjects = sc.maxSubjects
    c.gc_ticks = sc.gcTicks
    c.event_cap = sc.eventCap
    c.event_mask = sc.eventMask
    c.inbox_cap = sc.inboxCap
    c.device = sc.device
    c.shard_index = shard_index
    c.n_shards = n_shards
    c.target_scheme = sc.targetScheme
    c.join_pull = sc.joinPull
    c.pull_ticks = sc.pullTicks
    c.view_cap = sc.viewCap
    c.strict_reference_rules = 1 if sc.strictReferenceRules else 0
    c.push_pull = 1 if sc.pushPull else 0
    return c


class Sim:
    """All N members' `Store`s (src/Types.hs:53-60) behind one handle."""

    def __init__(self, abi, handle, sim_config: SimConfig):
        self._abi = abi
        self._h = handle
        self.simConfig = sim_config
        rc = _abi.Config()
        self._check(abi.get_config(handle, C.byref(rc)))
        self.resolved = rc
        self.nMembers = rc.n_members

    # -- lifecycle -------------------------------------------------------------
    @classmethod
    def create(cls, abi, sim_config: SimConfig, shard_index: int = 0, n_shards: int = 1) -> "Sim":
        err, sim = cls.configure(abi, sim_config, shard_index, n_shards)
        if err is not None:
            raise SwimError(-1, err)
        return sim

    @classmethod
    def configure(cls, abi, sim_config: SimConfig, shard_index: int = 0,
                  n_shards: int = 1) -> Tuple[Optional[str], Optional["Sim"]]:
        c = _to_c_config(sim_config, shard_index, n_shards)
        h = C.c_void_p()
        rc = abi.create(C.byref(c), C.byref(h))
        if rc != _abi.OK:
            msg = abi.last_error(None)
            return ((msg.decode() if msg else "error %d" % rc), None)
        return (None, cls(abi, h, sim_config))

    def close(self):
        if self._h:
            self._abi.destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc):
        if rc != _abi.OK:
            msg = self._abi.last_error(self._h)
            raise SwimError(rc, msg.decode() if msg else "")

    # -- fault injection ---------------------------------------------------------
    def scheduleFault(self, tick: int, member: int, up: bool):
        self._check(self._abi.schedule_fault(self._h, tick, member, 1 if up else 0))

    def crash(self, member: int, tick: int):
        self.scheduleFault(tick, member, False)

    # -- the hot path ------------------------------------------------------------
    def step(self, nticks: int = 1):
        """nticks periods of failureDetector for every member (src/Core.hs:236-240)."""
        self._check(self._abi.step(self._h, nticks))

    @property
    def tick(self) -> int:
        t = C.c_uint64()
        self._check(self._abi.tick(self._h, C.byref(t)))
        return t.value

    # -- results -----------------------------------------------------------------
    def drainEventsRaw(self) -> List[Tuple[int, int, int, int, int, int]]:
        """[(tick, observer, subject, incarnation, state, cause)] sorted."""
        n = C.c_size_t()
        rc = self._abi.drain_events(self._h, None, 0, C.byref(n))
        if rc == _abi.OK:
            return []
        if rc != _abi.ERR_BUFFER:
            self._check(rc)
        buf = (_abi.Event * n.value)()
        self._check(self._abi.drain_events(self._h, buf, n.value, C.byref(n)))
        return [(e.tick, e.observer, e.subject, e.incarnation, e.state, e.cause) for e in buf[: n.value]]

    def drainEventsArray(self):
        """The same records as one numpy structured array (fields tick, observer, subject, incarnation, state, cause), sorted as
        drainEventsRaw sorts them: for streams of millions of records (a million members with every cause recorded)."""
        import numpy as np
        dt = np.dtype([("tick", "<u8"), ("observer", "<u4"), ("subject", "<u4"), ("incarnation", "<u4"), ("state", "u1"), ("cause", "u1"), ("_pad", "<u2")])
        assert dt.itemsize == C.sizeof(_abi.Event)
        n = C.c_size_t()
        rc = self._abi.drain_events(self._h, None, 0, C.byref(n))
        if rc == _abi.OK:
            return np.zeros(0, dtype=dt)
        if rc != _abi.ERR_BUFFER:
            self._check(rc)
        out = np.zeros(n.value, dtype=dt)
        self._check(self._abi.drain_events(self._h, C.cast(out.ctypes.data, C.POINTER(_abi.Event)), n.value, C.byref(n)))
        out = out[: n.value]
        out["_pad"] = 0
        return out

    def drainEvents(self) -> List[MembershipEvent]:
        """The `Broadcast (Suspect|Alive|Dead ...)` gossip the members enqueued
        (src/Core.hs:119-121,254), in (tick, observer, subject) order."""
        return [MembershipEvent(t, memberName(o), Broadcast(event_message(st, inc, s, o)), cause)
                for (t, o, s, inc, st, cause) in self.drainEventsRaw()]

    def members(self, observer: int) -> List[Member]:
        """`members store` (src/Core.hs:76-77): the non-default entries of observer's map,
        name... id-sorted.  Every member not listed is Alive at incarnation 0."""
        n = C.c_size_t()
        rc = self._abi.read_view(self._h, observer, None, 0, C.byref(n))
        if rc == _abi.OK:
            return []
        if rc != _abi.ERR_BUFFER:
            self._check(rc)
        buf = (_abi.ViewEntry * n.value)()
        self._check(self._abi.read_view(self._h, observer, buf, n.value, C.byref(n)))
        return [Member(memberName(e.subject), Liveness(e.state), e.incarnation, e.since_tick)
                for e in buf[: n.value]]

    def readMember(self, member: int):
        m = _abi.Member()
        self._check(self._abi.read_member(self._h, member, C.byref(m)))
        rumors = sorted((r.subject, r.incarnation, r.state, r.tx_left) for r in m.rumors[: m.n_rumors])
        return {"id": m.id, "incarnation": m.incarnation, "up": bool(m.up), "n_timers": m.n_timers,
                "rumors": rumors}

    def firstDetection(self):
        """first tick at which a probe of j ended in Suspect while j was down, or None."""
        buf = (C.c_uint64 * self.nMembers)()
        self._check(self._abi.first_detect(self._h, buf, self.nMembers))
        return [None if v == _abi.TICK_NONE else v for v in buf]

    def firstDetectionArray(self):
        import numpy as np
        buf = np.empty(self.nMembers, dtype=np.uint64)
        self._check(self._abi.first_detect(self._h, buf.ctypes.data_as(C.POINTER(C.c_uint64)), self.nMembers))
        return buf

    def digest(self) -> int:
        d = C.c_uint64()
        self._check(self._abi.digest(self._h, C.byref(d)))
        return d.value

    def counters(self) -> dict:
        buf = (C.c_uint64 * _abi.CTR_COUNT)()
        self._check(self._abi.counters(self._h, buf, _abi.CTR_COUNT))
        return {name: buf[k] for k, name in enumerate(_abi.CTR_NAMES) if not name.startswith("_")}

    def tableStats(self) -> dict:
        """Occupancy of the bounded tables (product library only): view rows ever handed out / live subjects /
        reclaimed rows waiting / rumour ids handed out / view rows allocated / inbox overflow entries and room."""
        buf = (C.c_uint64 * 7)()
        self._check(self._abi.table_stats(self._h, buf, 7))
        return dict(zip(("rows_high_water", "subjects_live", "rows_free", "rumour_ids", "rows_allocated", "inbox_overflow",
                         "inbox_overflow_room"), buf))

    # -- measurement (product library only) -----------------------------------------
    def kernelTimingEnable(self, enable=True):
        self._check(self._abi.kernel_timing_enable(self._h, 1 if enable else 0))

    def kernelTiming(self):
        """{probe_ms, merge_ms, ticks}: HIP-event time spent in each tick kernel since enabling."""
        buf = (C.c_double * 3)()
        self._check(self._abi.kernel_timing(self._h, buf, 3))
        return {"probe_ms": buf[0], "merge_ms": buf[1], "ticks": int(buf[2])}

    # -- unit-level hooks (test/Spec.hs) -------------------------------------------
    def kRandomMembers(self, observer: int, n: int, excludes: Sequence[int] = ()) -> List[int]:
        """kRandomMembers store n excludes (src/Core.hs:69-74) for one observer."""
        ex = (C.c_uint32 * max(1, len(excludes)))(*excludes)
        out = (C.c_uint32 * max(1, n))()
        got = C.c_size_t()
        self._check(self._abi.k_random_members(self._h, observer, n, ex, len(excludes), out, n, C.byref(got)))
        return list(out[: got.value])

    def coverage(self, subject: int, state: int, incarnation: int = 0):
        """(holders, up): the up members other than `subject` whose entry about it is at least {incarnation, state} in the
        state rule's merge order, and the number of up members other than `subject` -- a rumour has reached everybody
        when the two are equal (BASELINE config 5: dissemination ticks-to-all)."""
        out = (C.c_uint64 * 2)()
        self._check(self._abi.coverage(self._h, subject, int(state), incarnation, out))
        return int(out[0]), int(out[1])

    def injectRumor(self, observer: int, subject: int, state: int, incarnation: int = 0):
        """A Suspect / Alive / Dead message about `subject` from OUTSIDE the simulation reaches `observer` in the next tick
        (`process` on a message off the socket, src/Core.hs:110-117; the live-node bridge uses it)."""
        self._check(self._abi.inject_rumor(self._h, observer, subject, int(state), incarnation))

    def setView(self, observer: int, subject: int, state: int, incarnation: int = 0):
        self._check(self._abi.set_view(self._h, observer, subject, int(state), incarnation))
