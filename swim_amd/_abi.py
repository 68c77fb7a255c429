"""ctypes declarations for the C ABI in include/swimsim.h.

`bind(lib, prefix)` attaches argtypes/restypes for every entry point of the header
under the given symbol prefix and returns a namespace of callables.  The product
uses prefix ``swimsim_`` on libswimsim.so (see _lib.py).  The parity tests reuse
the same declarations with prefix ``swimoracle_`` on the CPU oracle library --
that loading happens in tests/ only, never in this package.
"""
import ctypes as C

ABI_VERSION = 7

OK = 0
ERR_INVALID, ERR_DEVICE, ERR_NOMEM, ERR_CAPACITY, ERR_STATE, ERR_BUFFER = -1, -2, -3, -4, -5, -6
ALIVE, SUSPECT, DEAD = 0, 1, 2
TARGETS_RANDOM, TARGETS_ROBUST = 0, 1
CAUSE_PROBE, CAUSE_TIMER, CAUSE_GOSSIP, CAUSE_REFUTE, CAUSE_JOIN = 0, 1, 2, 3, 4
EVMASK_ALL = 0x1F
EVMASK_DEFAULT = (1 << CAUSE_PROBE) | (1 << CAUSE_REFUTE) | (1 << CAUSE_JOIN)
TICK_NONE = 0xFFFFFFFFFFFFFFFF
GC_AUTO = 0xFFFFFFFF

CTR_NAMES = [
    "pings", "direct_failed", "ping_reqs", "suspects", "false_suspects", "payloads",
    "rumors_seen", "changes", "pb_writes", "timers_fired", "refutes", "events_dropped",
    "active_members", "evdigest", "false_deads", "settled", "evicted",
]
CTR_COUNT = 17


class Config(C.Structure):
    """swimsim_config_t"""
    _fields_ = [
        ("struct_size", C.c_uint32), ("abi_version", C.c_uint32),
        ("num_to_gossip", C.c_int32), ("gossip_interval_us", C.c_int64),
        ("n_members", C.c_uint32), ("seed", C.c_uint64),
        ("probes_per_tick", C.c_int32), ("indirect_k", C.c_int32),
        ("loss_ppm", C.c_uint32), ("suspicion_ticks", C.c_uint32),
        ("retransmit_mult", C.c_uint32), ("max_subjects", C.c_uint32),
        ("gc_ticks", C.c_uint32), ("event_cap", C.c_uint32), ("event_mask", C.c_uint32),
        ("inbox_cap", C.c_uint32),
        ("device", C.c_int32), ("shard_index", C.c_uint32), ("n_shards", C.c_uint32),
        ("target_scheme", C.c_uint32), ("join_pull", C.c_uint32), ("pull_ticks", C.c_uint32),
        ("view_cap", C.c_uint32), ("strict_reference_rules", C.c_uint32), ("push_pull", C.c_uint32),
    ]


class Event(C.Structure):
    """swimsim_event_t"""
    _fields_ = [("tick", C.c_uint64), ("observer", C.c_uint32), ("subject", C.c_uint32),
                ("incarnation", C.c_uint32), ("state", C.c_uint8), ("cause", C.c_uint8),
                ("_pad", C.c_uint16)]


class ViewEntry(C.Structure):
    """swimsim_view_entry_t"""
    _fields_ = [("subject", C.c_uint32), ("incarnation", C.c_uint32), ("since_tick", C.c_uint32),
                ("state", C.c_uint8), ("_pad", C.c_uint8 * 3)]


class Rumor(C.Structure):
    """swimsim_rumor_t"""
    _fields_ = [("subject", C.c_uint32), ("incarnation", C.c_uint32), ("state", C.c_uint8),
                ("tx_left", C.c_uint8), ("_pad", C.c_uint16)]


class Member(C.Structure):
    """swimsim_member_t"""
    _fields_ = [("id", C.c_uint32), ("incarnation", C.c_uint32), ("up", C.c_uint8),
                ("n_rumors", C.c_uint8), ("n_timers", C.c_uint16), ("rumors", Rumor * 8)]


# name -> (restype, argtypes); the handle is an opaque void*
_H = C.c_void_p
_SIGS = {
    "default_config": (C.c_int, [C.POINTER(Config)]),
    "create": (C.c_int, [C.POINTER(Config), C.POINTER(_H)]),
    "create_msg": (C.c_int, [C.POINTER(Config), C.POINTER(_H), C.c_char_p, C.c_size_t]),
    "destroy": (None, [_H]),
    "last_error": (C.c_char_p, [_H]),
    "schedule_fault": (C.c_int, [_H, C.c_uint64, C.c_uint32, C.c_uint8]),
    "step": (C.c_int, [_H, C.c_uint32]),
    "tick": (C.c_int, [_H, C.POINTER(C.c_uint64)]),
    "drain_events": (C.c_int, [_H, C.POINTER(Event), C.c_size_t, C.POINTER(C.c_size_t)]),
    "read_view": (C.c_int, [_H, C.c_uint32, C.POINTER(ViewEntry), C.c_size_t, C.POINTER(C.c_size_t)]),
    "read_member": (C.c_int, [_H, C.c_uint32, C.POINTER(Member)]),
    "first_detect": (C.c_int, [_H, C.POINTER(C.c_uint64), C.c_size_t]),
    "digest": (C.c_int, [_H, C.POINTER(C.c_uint64)]),
    "counters": (C.c_int, [_H, C.POINTER(C.c_uint64), C.c_size_t]),
    "k_random_members": (C.c_int, [_H, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.c_size_t,
                                   C.POINTER(C.c_uint32), C.c_size_t, C.POINTER(C.c_size_t)]),
    "set_view": (C.c_int, [_H, C.c_uint32, C.c_uint32, C.c_uint8, C.c_uint32]),
    "get_config": (C.c_int, [_H, C.POINTER(Config)]),
    "inject_rumor": (C.c_int, [_H, C.c_uint32, C.c_uint32, C.c_uint8, C.c_uint32]),
    "coverage": (C.c_int, [_H, C.c_uint32, C.c_uint8, C.c_uint32, C.POINTER(C.c_uint64)]),
}

# entry points only the product library has (measurement plumbing, sharded stepping; no oracle counterpart)
_U32P = C.POINTER(C.c_uint32)
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32))   # swimsim_exchange_fn
_VPP = C.POINTER(C.c_void_p)
_PRODUCT_ONLY = {
    "shard_info": (C.c_int, [_H, _U32P, _U32P, _U32P, _U32P, _U32P]),
    "shard_buffers": (C.c_int, [_H, _VPP, _VPP]),
    "shard_phase1": (C.c_int, [_H, _U32P]),
    "shard_phase2": (C.c_int, [_H, _U32P, _U32P]),
    "shard_phase3": (C.c_int, [_H, _U32P, _U32P]),
    "shard_step": (C.c_int, [_H, C.c_uint32, C.c_void_p, C.c_void_p]),
    "cluster_step": (C.c_int, [C.POINTER(_H), C.c_uint32, C.c_uint32]),
    "shard_settle_buffers": (C.c_int, [_H, _VPP, _VPP, _U32P]),
    "shard_settle_counts": (C.c_int, [_H, _U32P]),
    "shard_settle_commit": (C.c_int, [_H, _U32P]),
    "shard_phase0": (C.c_int, [_H, _U32P, C.POINTER(C.c_int)]),
    "shard_gather_buffers": (C.c_int, [_H, _VPP, _VPP, _U32P]),
    "shard_join_buffers": (C.c_int, [_H, _VPP, _VPP, _U32P]),
    "shard_join_ingest": (C.c_int, [_H, _U32P]),
    "shard_traffic": (C.c_int, [_H, C.POINTER(C.c_uint64)]),
    "note_outside_rumor": (C.c_int, [_H, C.c_uint32, C.c_uint32]),
    "shard_get_first_suspect": (C.c_int, [_H, _U32P, C.c_size_t]),
    "shard_set_first_suspect": (C.c_int, [_H, _U32P, C.c_size_t]),
    "table_stats": (C.c_int, [_H, C.POINTER(C.c_uint64), C.c_size_t]),
    "kernel_timing_enable": (C.c_int, [_H, C.c_int]),
    "kernel_timing": (C.c_int, [_H, C.POINTER(C.c_double), C.c_size_t]),
}



class WireMsg(C.Structure):
    """swimwire_msg_t (include/swimwire.h)"""
    _fields_ = [("type", C.c_uint8), ("payload_len", C.c_uint8), ("port", C.c_uint16), ("seq_no", C.c_uint32),
                ("target", C.c_uint32), ("addr", C.c_uint32), ("incarnation", C.c_int64),
                ("node", C.c_char * 64), ("dead_from", C.c_char * 64), ("payload", C.c_uint8 * 255)]


# include/swimwire.h: the wire codec (prefix swimwire_, product library only)
_WIRE = {
    "encode": (C.c_int, [C.POINTER(WireMsg), C.c_size_t, C.POINTER(C.c_uint8), C.c_size_t, C.POINTER(C.c_size_t)]),
    "decode": (C.c_int, [C.POINTER(C.c_uint8), C.c_size_t, C.POINTER(WireMsg), C.c_size_t, C.POINTER(C.c_size_t)]),
    "size": (C.c_int, [C.POINTER(WireMsg), C.c_size_t, C.POINTER(C.c_size_t)]),
    "encode_bare": (C.c_int, [C.POINTER(WireMsg), C.POINTER(C.c_uint8), C.c_size_t, C.POINTER(C.c_size_t)]),
    "decode_any": (C.c_int, [C.POINTER(C.c_uint8), C.c_size_t, C.POINTER(WireMsg), C.c_size_t, C.POINTER(C.c_size_t),
                             C.POINTER(C.c_int)]),
    "last_error": (C.c_char_p, []),
}

class BridgeStats(C.Structure):
    """swimbridge_stats_t (include/swimbridge.h)"""
    _fields_ = [(n, C.c_uint64) for n in ("datagrams_in", "datagrams_out", "decode_errors", "pings", "pings_unanswered",
                                          "indirect_pings", "relayed_acks", "acks_in", "rumors_injected", "rumors_foreign",
                                          "rumors_dropped", "sends_failed", "bare_in")]


# include/swimbridge.h: the live-node bridge (prefix swimbridge_, product library only)
_BRIDGE = {
    "open": (C.c_int, [_H, C.c_char_p, C.c_uint16, C.POINTER(C.c_void_p)]),
    "open_cluster": (C.c_int, [C.POINTER(C.c_void_p), C.c_uint32, C.c_char_p, C.c_uint16, C.POINTER(C.c_void_p)]),
    "port": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint16)]),
    "poll": (C.c_int, [C.c_void_p, C.c_int, C.c_uint32]),
    "stats": (C.c_int, [C.c_void_p, C.POINTER(BridgeStats)]),
    "accept_bare": (C.c_int, [C.c_void_p, C.c_int]),
    "last_error": (C.c_char_p, [C.c_void_p]),
    "close": (None, [C.c_void_p]),
}

ENTRY_POINTS = tuple(_SIGS) + tuple(_PRODUCT_ONLY)
BRIDGE_ENTRY_POINTS = tuple("swimbridge_" + n for n in _BRIDGE)
WIRE_ENTRY_POINTS = tuple("swimwire_" + n for n in _WIRE)


class Namespace:
    pass


def bind(lib, prefix):
    """Resolve every entry point of swimsim.h as `prefix + name` in `lib`.

    Raises AttributeError naming the first missing symbol (fail loudly)."""
    ns = Namespace()
    sigs = dict(_SIGS)
    if prefix == "swimsim_":
        sigs.update(_PRODUCT_ONLY)
    for name, (res, args) in sigs.items():
        fn = getattr(lib, prefix + name)
        fn.restype = res
        fn.argtypes = args
        setattr(ns, name, fn)
    if prefix == "swimsim_":
        for name, (res, args) in _WIRE.items():
            fn = getattr(lib, "swimwire_" + name)
            fn.restype = res
            fn.argtypes = args
            setattr(ns, "wire_" + name, fn)
        for name, (res, args) in _BRIDGE.items():
            fn = getattr(lib, "swimbridge_" + name)
            fn.restype = res
            fn.argtypes = args
            setattr(ns, "bridge_" + name, fn)
    ns.lib = lib
    ns.prefix = prefix
    return ns
