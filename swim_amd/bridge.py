"""The live-node bridge (include/swimbridge.h): one UDP endpoint behind which the simulated members
answer the reference's wire protocol -- `handleUDPMessage.process` (src/Core.hs:79-117) for datagrams from outside.

    sim = Sim.create(abi, sc)
    with Bridge(sim, port=4000) as br:         # the reference binds 127.0.0.1:4000 (src/Core.hs:278)
        while running:
            br.poll(timeout_ms=10)              # answer Pings / IndirectPings, queue the gossip that came in
            sim.step(1)                         # one protocol period for the whole population

Host-only code of libswimsim.so (POSIX sockets + the wire codec); this module is the ctypes mirror."""
import ctypes as C

from . import _abi


class Bridge:
    def __init__(self, sim, bind_ip: str = "127.0.0.1", port: int = 0):
        """sim: a Sim, or a ShardedSim whose shards all live in this process (swimbridge_open_cluster: one endpoint for the cluster)."""
        self._b = C.c_void_p()
        if hasattr(sim, "shards"):
            if len(sim.shards) != sim.n_shards:
                raise OSError("the bridge of a cluster needs every shard in this process")
            self._abi = sim.shards[0].sim._abi
            arr = (C.c_void_p * sim.n_shards)(*[s.sim._h for s in sim.shards])
            rc = self._abi.bridge_open_cluster(arr, sim.n_shards, bind_ip.encode(), port, C.byref(self._b))
        else:
            self._abi = sim._abi
            rc = self._abi.bridge_open(sim._h, bind_ip.encode(), port, C.byref(self._b))
        if rc:
            raise OSError("swimbridge_open(%s:%d) failed with status %d" % (bind_ip, port, rc))
        self._sim = sim

    @property
    def port(self) -> int:
        p = C.c_uint16()
        self._abi.bridge_port(self._b, C.byref(p))
        return p.value

    def poll(self, timeout_ms: int = 0, max_datagrams: int = 1024) -> int:
        """Handle the datagrams that are waiting; returns how many."""
        n = self._abi.bridge_poll(self._b, timeout_ms, max_datagrams)
        if n < 0:
            raise OSError("swimbridge_poll: status %d: %s" % (n, (self._abi.bridge_last_error(self._b) or b"").decode()))
        return n

    def acceptBare(self, on: bool = True):
        """Also accept the bare-`Message` datagrams a LITERAL reference node sends (D11; src/Core.hs:133-134)."""
        self._abi.bridge_accept_bare(self._b, 1 if on else 0)

    def stats(self) -> dict:
        st = _abi.BridgeStats()
        self._abi.bridge_stats(self._b, C.byref(st))
        return {n: getattr(st, n) for n, _ in _abi.BridgeStats._fields_}

    def close(self):
        if self._b:
            self._abi.bridge_close(self._b)
            self._b = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
