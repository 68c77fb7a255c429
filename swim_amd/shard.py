"""ShardedSim: one simulated population split over several handles (one per GPU).

Members shard by contiguous id range; every shard holds the same configuration and the same
fault schedule.  Probe outcomes need no communication (ground truth and the loss hashes are
known everywhere); only piggyback payloads cross shards, in two rounds per tick
(include/swimsim.h, "sharded clusters"):

    phase1  begin + publish         -> round 1: ALL-GATHER of every shard's ring dictionary (+ the queues that travel
                                       as lists), queue masks (8 B / member) and queue bytes (1 B / member)
    phase2  xlat + probe            -> round 2: 8-byte records {dst, src} "dst merges src's queue" to the owner of dst
    phase3  ingest + merge

Two fabrics move the records:
  * LocalFabric  -- all shards live in this process (several handles on one device): plain copies.
                    This is how the sharded path is checked against the unsharded one on a single GPU.
  * DistFabric   -- one shard per process, `torch.distributed` point-to-point sends
                    (backend "nccl" = RCCL over xGMI on GPUs, "gloo" for the CPU tests).
PyTorch is only plumbing here (device buffers are wrapped zero-copy); all compute is in libswimsim.so.
"""
import ctypes as C
from typing import List, Sequence

from . import _abi
from .sim import Sim, SwimError
from .types import SimConfig

_M64 = (1 << 64) - 1
REC_BYTES = (16, 8, 72, 8, 16, 8, 1)    # record kinds: round-1 records (dictionary + lists: ONE segment for every peer), round-2
                                        # records {dst, src}, (none any more), settle records, join pulls, replicated queue masks,
                                        # replicated queue bytes (all-gathered)


class _NoView:
    """Stands in for a buffer view that could not be made (torch refused the pointer on this device): a cluster stepped inside the
    library (swimsim_cluster_step) never needs the views; the embedder-side exchange says so when it does."""

    def __init__(self, why):
        self.why = why

    def __getattr__(self, name):
        raise RuntimeError("no torch view of the library's exchange buffer: %s" % self.why)

    def __getitem__(self, k):
        raise RuntimeError("no torch view of the library's exchange buffer: %s" % self.why)


def _wrap(ptr: int, nbytes: int, device):
    """Zero-copy uint8 torch view of library-owned memory (device or host)."""
    try:
        return _wrap_now(ptr, nbytes, device)
    except Exception as e:          # noqa: BLE001 -- plumbing for ONE of the two ways to step a cluster; the other does without
        class _Flat(_NoView):
            def view(self, *a):
                return self

            def expand(self, *a):
                return self
        return _Flat(repr(e)[:200])


def _wrap_now(ptr: int, nbytes: int, device):
    import torch
    if nbytes == 0:
        return torch.empty(0, dtype=torch.uint8, device=device)
    if str(device).startswith("cuda"):
        class _Iface:
            __cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}
        return torch.as_tensor(_Iface(), device=device)
    import numpy as np
    buf = (C.c_uint8 * nbytes).from_address(ptr)
    return torch.from_numpy(np.ctypeslib.as_array(buf))


class _Shard:
    """One handle + views of its exchange buffers (send[kind][peer, bytes], recv[kind][peer, bytes])."""

    def __init__(self, abi, sim_config: SimConfig, index: int, n_shards: int, device):
        self.index, self.n_shards = index, n_shards
        self.sim = Sim.create(abi, sim_config, shard_index=index, n_shards=n_shards)
        a, h, G = abi, self.sim._h, n_shards
        # record sizes by kind; bounded handles (view_cap) all-gather whole 64-byte queue lines where dense ones gather 8-byte masks
        self.rec_bytes = list(REC_BYTES)
        bounded = bool(self.sim.resolved.view_cap)
        if bounded:
            self.rec_bytes[5] = 64
            self.rec_bytes[1] = 16
        REC = self.rec_bytes
        # kinds whose send "segments" are ONE slice that goes to every peer
        self.shared_kinds = (5, 6) if bounded else (0, 5, 6)
        vals = [C.c_uint32() for _ in range(5)]
        self.sim._check(a.shard_info(h, *[C.byref(v) for v in vals]))
        self.lo, self.n_local = vals[0].value, vals[1].value
        caps = [v.value for v in vals[2:]]
        sp, rp = (C.c_void_p * 3)(), (C.c_void_p * 3)()
        self.sim._check(a.shard_buffers(h, sp, rp))
        def seg(ptr, k, shared):
            nb = caps[k] * REC[k] if ptr else 0
            if shared:
                return _wrap(ptr or 0, nb, device).view(1, nb).expand(G, nb)
            return _wrap(ptr or 0, G * nb, device).view(G, nb)
        self.send = [seg(sp[k], k, k in self.shared_kinds) for k in range(3)]
        self.recv = [seg(rp[k], k, False) for k in range(3)]
        self.settling = self.sim.resolved.gc_ticks != 0
        self.join_pull = self.sim.resolved.join_pull != 0 or self.sim.resolved.pull_ticks != 0   # state pulls: exchange round 0 (kind 4)
        # kind 3: what every shard says about its rows (round 3, settling); kind 4: join-time pulls (round 0)
        for kind, on, fn in ((3, self.settling, a.shard_settle_buffers), (4, self.join_pull, a.shard_join_buffers)):
            if not on:
                self.send.append(None); self.recv.append(None)
                continue
            sp_, rp_, cp_ = C.c_void_p(), C.c_void_p(), C.c_uint32()
            self.sim._check(fn(h, C.byref(sp_), C.byref(rp_), C.byref(cp_)))
            nb = cp_.value * REC[kind]
            self.send.append(_wrap(sp_.value, G * nb, device).view(G, nb))
            self.recv.append(_wrap(rp_.value, G * nb, device).view(G, nb))

        # kinds 5 / 6: the replicated queue masks / bytes (all-gather after phase 1): the send "segments" of all
        # peers are one and the same slice, the receive segments are the peers' slices of the whole table
        gs, gr, gn = (C.c_void_p * 2)(), (C.c_void_p * 2)(), C.c_uint32()
        self.sim._check(a.shard_gather_buffers(h, gs, gr, C.byref(gn)))
        self.replicated = gn.value != 0
        for k in (0, 1):
            if not self.replicated:
                self.send.append(None); self.recv.append(None)
                continue
            nb = gn.value * REC[5 + k]
            self.send.append(_wrap(gs[k], nb, device).view(1, nb).expand(G, nb))
            self.recv.append(_wrap(gr[k], G * nb, device).view(G, nb))

    def phase0(self):
        c, need = (C.c_uint32 * self.n_shards)(), C.c_int()
        self.sim._check(self.sim._abi.shard_phase0(self.sim._h, c, C.byref(need)))
        return list(c), bool(need.value)

    def join_ingest(self, j_in: Sequence[int]):
        self.sim._check(self.sim._abi.shard_join_ingest(self.sim._h, (C.c_uint32 * self.n_shards)(*j_in)))

    def phase1(self):
        G = self.n_shards
        c = (C.c_uint32 * (3 * G))()
        self.sim._check(self.sim._abi.shard_phase1(self.sim._h, c))
        return [list(c[k * G:(k + 1) * G]) for k in range(3)]

    def phase2(self, r_in: Sequence[int]):
        G = self.n_shards
        c = (C.c_uint32 * (3 * G))()
        self.sim._check(self.sim._abi.shard_phase2(self.sim._h, (C.c_uint32 * G)(*r_in), c))
        return [list(c[k * G:(k + 1) * G]) for k in range(3)]

    def phase3(self, p_in: Sequence[int], x_in: Sequence[int]):
        G = self.n_shards
        self.sim._check(self.sim._abi.shard_phase3(self.sim._h, (C.c_uint32 * G)(*p_in), (C.c_uint32 * G)(*x_in)))

    def settle_counts(self):
        c = (C.c_uint32 * self.n_shards)()
        self.sim._check(self.sim._abi.shard_settle_counts(self.sim._h, c))
        return list(c)

    def settle_commit(self, s_in: Sequence[int]):
        self.sim._check(self.sim._abi.shard_settle_commit(self.sim._h, (C.c_uint32 * self.n_shards)(*s_in)))


class LocalFabric:
    """All shards in this process: record exchange = copies between their buffers."""

    def __init__(self, n_shards):
        self.n_shards = n_shards
        self.local = list(range(n_shards))
        self.rank0 = True

    def exchange(self, shards: List[_Shard], kinds, counts):
        """counts[s][j][p] = records of kind kinds[j] local shard s sends to shard p.
        Returns recv[s][j][p] = records shard s received from p."""
        G = self.n_shards
        recv = [[[0] * G for _ in kinds] for _ in shards]
        for si, src in enumerate(shards):
            for j, kind in enumerate(kinds):
                for p in range(G):
                    n = counts[si][j][p]
                    if n == 0:
                        continue
                    assert p != src.index, "a shard never sends to itself"
                    nb = n * src.rec_bytes[kind]
                    shards[p].recv[kind][src.index, :nb].copy_(src.send[kind][0 if kind in src.shared_kinds else p, :nb])
                    recv[p][j][src.index] = n
        if shards and shards[0].send[0].is_cuda:
            import torch
            torch.cuda.synchronize()                # torch copies are asynchronous; the library has its own stream
        return recv

    def gather(self, obj):
        return [obj]

    def reduce_min_u32(self, arrays):
        import numpy as np
        out = arrays[0]
        for a in arrays[1:]:
            out = np.minimum(out, a)
        return out


class DistFabric:
    """One shard per process; torch.distributed moves the records.

    transport "nccl": device-to-device sends (RCCL over xGMI), records staged through torch-allocated
    tensors so that RCCL only ever sees memory from torch's allocator.  transport "host": records are
    copied to host memory and sent with the CPU backend (gloo) -- the CPU tests, and the fallback when a
    point-to-point self-test over RCCL fails on the machine at hand (the choice is reported, never silent).
    """

    def __init__(self, device, transport="auto"):
        import torch
        import torch.distributed as dist
        self.dist, self.torch = dist, torch
        self.n_shards = dist.get_world_size()
        self.rank = dist.get_rank()
        self.local = [self.rank]
        self.rank0 = self.rank == 0
        self.device = device
        self.on_gpu = str(device).startswith("cuda")
        self.note = ""
        if not self.on_gpu:
            transport = "host"
        elif transport == "auto":
            transport, self.note = self._self_test()
        self.transport = transport

    def _self_test(self):
        """Ring send/recv of a small device tensor over the device backend; all ranks agree on the result."""
        torch, dist = self.torch, self.dist
        ok, why = 1, ""
        try:
            if self.n_shards > 1:
                a = torch.full((256,), self.rank, dtype=torch.uint8, device=self.device)
                b = torch.empty_like(a)
                nxt, prv = (self.rank + 1) % self.n_shards, (self.rank - 1) % self.n_shards
                for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, a, nxt), dist.P2POp(dist.irecv, b, prv)]):
                    w.wait()
                torch.cuda.synchronize()
                ok = int(int(b[0].item()) == prv)
        except Exception as e:                      # noqa: BLE001 -- reported in the bench line
            ok, why = 0, repr(e)[:200]
        flag = torch.tensor([ok], dtype=torch.int64)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self._cpu_group())
        if int(flag.item()):
            return "nccl", ""
        return "host", "device p2p self-test failed (%s): records staged through host memory" % (why or "peer")

    def _cpu_group(self):
        if not hasattr(self, "_cpug"):
            self._cpug = self.dist.new_group(backend="gloo")
        return self._cpug

    def _staging(self, sh, host):
        """Persistent torch-allocated staging buffers (send, recv) shaped like the library's, so that the
        collective backend only sees torch memory and a tick allocates nothing."""
        if not hasattr(self, "_stage"):
            torch = self.torch
            dev = "cpu" if host else self.device
            mk = lambda bufs, one: [None if b is None else torch.empty((1, b.shape[1]) if (one and k in sh.shared_kinds) else tuple(b.shape),
                                                                       dtype=torch.uint8, device=dev, pin_memory=(host and self.on_gpu))
                                    for k, b in enumerate(bufs)]
            self._stage = (mk(sh.send, True), mk(sh.recv, False))
        return self._stage

    def exchange(self, shards, kinds, counts):
        """One round: the counts of every kind in ONE all_to_all, then all records in ONE batch of p2p ops."""
        torch, dist, G, me = self.torch, self.dist, self.n_shards, self.rank
        sh, nk = shards[0], len(kinds)
        host = self.transport == "host"
        grp = self._cpu_group() if (host and self.on_gpu) else None
        # kinds 5 / 6 = the all-gather of the replicated tables: every count is n_local, nothing to ask the peers
        vk = [j for j, k in enumerate(kinds) if k not in (5, 6)]
        recv = [[0 if p == me else sh.n_local for p in range(G)] for _ in kinds]
        if vk:
            flat = [counts[0][j][p] for p in range(G) for j in vk]             # peer-major for all_to_all
            cs = torch.tensor(flat, dtype=torch.int64, device="cpu" if host else self.device)
            cr = torch.empty_like(cs)
            dist.all_to_all_single(cr, cs, group=grp)
            got = [int(v) for v in cr.tolist()]
            for x, j in enumerate(vk):
                recv[j] = [got[p * len(vk) + x] for p in range(G)]
        ops, landing = [], []
        stage = self._staging(sh, host)
        staged = set()
        for p in range(G):
            if p == me:
                continue
            for j, kind in enumerate(kinds):
                n_out, n_in = counts[0][j][p], recv[j][p]
                if n_out:
                    nb = n_out * sh.rec_bytes[kind]
                    row = 0 if kind in sh.shared_kinds else p   # the same slice goes to every peer: staged once
                    out = stage[0][kind][row, :nb]
                    if (kind, row) not in staged:
                        out.copy_(sh.send[kind][row, :nb])
                        staged.add((kind, row))
                    ops.append(dist.P2POp(dist.isend, out, p, group=grp))
                if n_in:
                    tmp = stage[1][kind][p, : n_in * sh.rec_bytes[kind]]
                    landing.append((kind, p, tmp))
                    ops.append(dist.P2POp(dist.irecv, tmp, p, group=grp))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        for kind, p, tmp in landing:
            sh.recv[kind][p, : tmp.numel()].copy_(tmp)
        if self.on_gpu:
            torch.cuda.synchronize()                # the library launches on its own stream
        return [recv]

    def gather(self, obj):
        out = [None] * self.n_shards
        self.dist.all_gather_object(out, obj, group=self._cpu_group() if self.on_gpu else None)
        return out

    def reduce_min_u32(self, arrays):
        import numpy as np
        t = self.torch.from_numpy(arrays[0].astype(np.int64))
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN, group=self._cpu_group() if self.on_gpu else None)
        return t.numpy().astype(np.uint32)


class ShardedSim:
    """The whole population behind one object, whatever the number of shards / processes."""

    def __init__(self, abi, sim_config: SimConfig, fabric, device="cpu", devices=None):
        """devices: one HIP device ordinal per LOCAL shard (a one-process cluster over the GPUs of a node: shard k on GPU
        devices[k], stepped by swimsim_cluster_step); default: every shard on sim_config.device, buffers wrapped for `device`."""
        import copy
        self.fabric = fabric
        self.n_shards = fabric.n_shards
        self.simConfig = sim_config
        self.shards = []
        for j, k in enumerate(fabric.local):
            sc_k, dev_k = sim_config, device
            if devices is not None:
                sc_k = copy.copy(sim_config); sc_k.device = devices[j]; dev_k = "cuda:%d" % devices[j]
            self.shards.append(_Shard(abi, sc_k, k, self.n_shards, dev_k))
        self.nMembers = sim_config.nMembers
        self.n_local = self.shards[0].n_local
        self.resolved = self.shards[0].sim.resolved
        self._fd_synced_at = -1
        self._injected = False
        import os
        self._in_library = os.environ.get("SWIMSIM_CLUSTER_STEP", "1") != "0"   # read when the cluster is made (tests force the phase calls)
        self.phase_seconds = [0.0] * 5          # host-side wall time per step part: phase1, round1, phase2, round2, phase3
        self.timed_ticks = 0

    def close(self):
        for s in self.shards:
            s.sim.close()

    # -- fault injection: every shard gets the whole schedule (ground truth is replicated) ------------
    def scheduleFault(self, tick: int, member: int, up: bool):
        for s in self.shards:
            s.sim.scheduleFault(tick, member, up)

    def crash(self, member: int, tick: int):
        self.scheduleFault(tick, member, False)

    def injectRumor(self, observer: int, subject: int, state: int, incarnation: int = 0):
        """A message from outside the simulation (swimsim_inject_rumor) goes to the shard that owns the observer; in a cluster of one
        process per shard every rank makes the call, the owner's takes it."""
        s = self._owner(observer)
        self._injected = True
        if s is not None:
            s.sim.injectRumor(observer, subject, state, incarnation)
        for o in self.shards:                         # the other shards open the subject's view row (a row belongs to the whole cluster)
            if o is not s:
                o.sim._check(o.sim._abi.note_outside_rumor(o.sim._h, observer, subject))

    # -- the hot path ------------------------------------------------------------------------------------
    def _step_by_library(self, nticks: int):
        """One shard per process: the tick loop runs inside libswimsim.so (swimsim_shard_step); this
        process only lends it the all-to-all-v -- exactly what a host without this module would bind."""
        import time
        sh, f, G = self.shards[0], self.fabric, self.n_shards
        acc = self.phase_seconds
        state = {"t": time.perf_counter()}

        def xchg(_ctx, rnd, c_out, c_in):
            try:
                t0 = time.perf_counter()
                acc[{0: 0, 1: 0, 2: 2, 3: 4}[rnd]] += t0 - state["t"]  # the phase that just ended
                kinds = {0: (4,), 1: (0, 5, 6) if sh.replicated else (0,), 2: (1,), 3: (3,)}[rnd]
                # rounds 0 and 3: one kind, its counts at [p]; round 1: kind 0 at [p], the gathered kinds 5 and 6 at [G + p], [2G + p]
                at = (lambda k: 0) if rnd in (0, 3) else (lambda k: (k - 4) * G if k >= 5 else k * G)
                counts = [[[c_out[at(k) + p] for p in range(G)] for k in kinds]]
                got = f.exchange([sh], kinds, counts)[0]
                for j, k in enumerate(kinds):
                    for p in range(G):
                        c_in[at(k) + p] = got[j][p]
                state["t"] = time.perf_counter()
                acc[{0: 1, 1: 1, 2: 3, 3: 4}[rnd]] += state["t"] - t0
                return 0
            except Exception:                                       # noqa: BLE001 -- must not unwind through C
                import traceback
                traceback.print_exc()
                return 1
        cb = _abi.EXCHANGE_FN(xchg)
        for _ in range(nticks):
            state["t"] = time.perf_counter()
            sh.sim._check(sh.sim._abi.shard_step(sh.sim._h, 1, cb, None))
            acc[4] += time.perf_counter() - state["t"]
            self.timed_ticks += 1

    def step(self, nticks: int = 1):
        f, sh = self.fabric, self.shards
        if len(sh) == 1 and self.n_shards > 1:
            return self._step_by_library(nticks)
        import os
        in_library = True
        self._injected = False
        if len(sh) == self.n_shards > 1 and in_library and self._in_library:
            # every shard of the cluster lives in this process: the library steps the cluster itself, the
            # exchange enqueued on the handles' streams (swimsim_cluster_step: no host in the loop)
            import time
            t0 = time.perf_counter()
            arr = (C.c_void_p * self.n_shards)(*[s.sim._h for s in sh])
            rc = sh[0].sim._abi.cluster_step(arr, self.n_shards, nticks)
            if rc != _abi.OK:
                bad = next((s for s in sh if (s.sim._abi.last_error(s.sim._h) or b"")), sh[0])
                raise SwimError(rc, (bad.sim._abi.last_error(bad.sim._h) or b"").decode())
            self.phase_seconds[4] += time.perf_counter() - t0
            self.timed_ticks += nticks
            return
        import time
        acc = self.phase_seconds
        for _ in range(nticks):
            t0 = time.perf_counter()
            if sh[0].join_pull:                                                     # round 0: state pulls (join-time, periodic)
                c0 = [s.phase0() for s in sh]
                if c0[0][1]:
                    j_in = f.exchange(sh, (4,), [[c[0]] for c in c0])
                    for k, s in enumerate(sh):
                        s.join_ingest(j_in[k][0])
            c1 = [s.phase1() for s in sh]
            t1 = time.perf_counter()
            if sh[0].replicated:                                                    # round 1 + the all-gather of the queue masks
                full = lambda s: [0 if p == s.index else s.n_local for p in range(self.n_shards)]
                r_in = f.exchange(sh, (0, 5, 6), [[c[0], full(s), full(s)] for c, s in zip(c1, sh)])
            else:
                r_in = f.exchange(sh, (0,), [[c[0]] for c in c1])                   # round 1
            t2 = time.perf_counter()
            c2 = [s.phase2(r_in[k][0]) for k, s in enumerate(sh)]
            t3 = time.perf_counter()
            px_in = f.exchange(sh, (1,), [[c[1]] for c in c2])                      # round 2
            t4 = time.perf_counter()
            for k, s in enumerate(sh):
                s.phase3(px_in[k][0], [0] * self.n_shards)
            if sh[0].settling:                                                      # round 3: settle records
                s_in = f.exchange(sh, (3,), [[s.settle_counts()] for s in sh])
                for k, s in enumerate(sh):
                    s.settle_commit(s_in[k][0])
            t5 = time.perf_counter()
            for j, d in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
                acc[j] += d
            self.timed_ticks += 1

    def phaseBreakdown(self, reset=False):
        """Mean host-side wall time (us) per tick of phase1 / round 1 / phase2 / round 2 / phase3 since the last reset."""
        n = max(1, self.timed_ticks)
        out = {k: round(v / n * 1e6, 1) for k, v in zip(("phase1", "round1", "phase2", "round2", "phase3"), self.phase_seconds)}
        if reset:
            self.phase_seconds = [0.0] * 5
            self.timed_ticks = 0
        return out

    @property
    def tick(self) -> int:
        return self.shards[0].sim.tick

    # -- results --------------------------------------------------------------------------------------------
    def _sync_first_suspect(self):
        """First-detection ticks are recorded by the prober's shard: combine (min) and set back."""
        import numpy as np
        if self._fd_synced_at == self.tick:
            return
        n = self.nMembers
        arrays = []
        for s in self.shards:
            buf = np.empty(n, dtype=np.uint32)
            s.sim._check(s.sim._abi.shard_get_first_suspect(s.sim._h, buf.ctypes.data_as(C.POINTER(C.c_uint32)), n))
            arrays.append(buf)
        comb = np.ascontiguousarray(self.fabric.reduce_min_u32(arrays))
        for s in self.shards:
            s.sim._check(s.sim._abi.shard_set_first_suspect(s.sim._h, comb.ctypes.data_as(C.POINTER(C.c_uint32)), n))
        self._fd_synced_at = self.tick

    def counters(self) -> dict:
        local = [s.sim.counters() for s in self.shards]
        parts = [c for group in self.fabric.gather(local) for c in group]
        return {k: sum(p[k] for p in parts) & _M64 for k in parts[0]}

    def digest(self) -> int:
        self._sync_first_suspect()
        local = sum(s.sim.digest() for s in self.shards) & _M64
        return sum(self.fabric.gather(local)) & _M64

    def coverage(self, subject: int, state: int, incarnation: int = 0):
        """(holders, up) over the whole population: every shard counts the members it owns (collective)."""
        local = [s.sim.coverage(subject, state, incarnation) for s in self.shards]
        parts = [c for group in self.fabric.gather(local) for c in group]
        return sum(p[0] for p in parts), sum(p[1] for p in parts)

    def drainEventsRaw(self):
        local = [e for s in self.shards for e in s.sim.drainEventsRaw()]
        return sorted(e for group in self.fabric.gather(local) for e in group)

    def firstDetection(self):
        self._sync_first_suspect()
        return self.shards[0].sim.firstDetection()

    def _owner(self, member: int):
        for s in self.shards:
            if s.lo <= member < s.lo + s.n_local:
                return s
        return None

    def members(self, observer: int):
        s = self._owner(observer)
        local = s.sim.members(observer) if s is not None else None
        return next(v for v in self.fabric.gather(local) if v is not None)

    def readMember(self, member: int):
        s = self._owner(member)
        local = s.sim.readMember(member) if s is not None else None
        return next(v for v in self.fabric.gather(local) if v is not None)

    def kernelTimingEnable(self, enable=True):
        for s in self.shards:
            s.sim.kernelTimingEnable(enable)

    def kernelTiming(self):
        return self.shards[0].sim.kernelTiming()

    def traffic(self):
        """What each local shard put on the wire in the last tick (swimsim_shard_traffic): a list of dicts."""
        out = []
        for s in self.shards:
            v = (C.c_uint64 * 4)()
            s.sim._check(s.sim._abi.shard_traffic(s.sim._h, v))
            out.append({"round1_bytes_to_each_peer": int(v[0]), "round2_bytes_to_all_peers": int(v[1]), "round2_records_kept": int(v[2]),
                        "queues_as_lists": int(v[3])})
        return out
