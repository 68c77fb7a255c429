"""The C-ABI library loads and exports every symbol include/swimsim.h declares (no GPU needed:
no compute entry point is called), and the product never falls back to a CPU path."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "swimsim.h")).read()
    return sorted(set(re.findall(r"\b(swimsim_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    lib = C.CDLL(os.path.join(ROOT, "swim_amd", "csrc", "libswimsim.so"))
    syms = header_symbols()
    assert "swimsim_step" in syms and "swimsim_create" in syms and len(syms) >= 18
    for name in syms:
        assert hasattr(lib, name), "missing export: " + name


def test_wire_codec_symbols_are_exported_and_bound():
    """include/swimwire.h: every declared entry point is exported by the same library and bound in _abi."""
    import __graft_entry__ as g
    from swim_amd import _abi
    g.build()
    text = open(os.path.join(ROOT, "include", "swimwire.h")).read()
    syms = sorted(set(re.findall(r"\b(swimwire_[a-z_0-9]+)\s*\(", text)))
    lib = C.CDLL(os.path.join(ROOT, "swim_amd", "csrc", "libswimsim.so"))
    assert syms == sorted(_abi.WIRE_ENTRY_POINTS)
    for name in syms:
        assert hasattr(lib, name), "missing export: " + name
    src = '#include <stdio.h>\n#include "swimwire.h"\nint main(){printf("%zu\\n", sizeof(swimwire_msg_t));return 0;}'
    import subprocess, tempfile
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "t.c"), "-o", os.path.join(d, "t")])
        assert int(subprocess.check_output([os.path.join(d, "t")]).split()[0]) == C.sizeof(_abi.WireMsg)


def test_generated_haskell_offsets_are_current():
    """haskell/Swim/Offsets.hs is generated from the headers (scripts/gen_hs_offsets.py); a stale file fails
    here, so the uncompiled shim cannot drift from the ABI."""
    import subprocess, sys
    assert subprocess.call([sys.executable, os.path.join(ROOT, "scripts", "gen_hs_offsets.py"), "--check"]) == 0
    shim = open(os.path.join(ROOT, "haskell", "Swim", "Sim.hs")).read()
    assert not re.search(r"(peek|poke)ByteOff \w+ \d", shim), "hand-typed offset in the Haskell shim"
    for name in re.findall(r"\bswim(?:sim|wire)[A-Z]\w*|\bcSwim\w+", shim):
        assert name in open(os.path.join(ROOT, "haskell", "Swim", "Offsets.hs")).read(), name


def test_bridge_symbols_are_exported_and_bound():
    """include/swimbridge.h: every declared entry point is exported by the same library and bound in _abi."""
    import __graft_entry__ as g
    from swim_amd import _abi
    g.build()
    text = open(os.path.join(ROOT, "include", "swimbridge.h")).read()
    syms = sorted(set(re.findall(r"\b(swimbridge_[a-z_0-9]+)\s*\(", text)))
    lib = C.CDLL(os.path.join(ROOT, "swim_amd", "csrc", "libswimsim.so"))
    assert syms == sorted(_abi.BRIDGE_ENTRY_POINTS)
    for name in syms:
        assert hasattr(lib, name), "missing export: " + name


def test_python_binding_covers_the_header():
    from swim_amd import _abi
    declared = {s[len("swimsim_"):] for s in header_symbols()}
    assert declared == set(_abi.ENTRY_POINTS)


def test_config_struct_layout_matches_header():
    """sizeof/offset check of swimsim_config_t against a tiny C program's view (gcc)."""
    import subprocess, tempfile
    from swim_amd import _abi
    src = '#include <stdio.h>\n#include <stddef.h>\n#include "swimsim.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(swimsim_config_t), offsetof(swimsim_config_t, seed), offsetof(swimsim_config_t, inbox_cap), offsetof(swimsim_config_t, n_shards), sizeof(swimsim_event_t), sizeof(swimsim_member_t));return 0;}'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "t.c"), "-o", os.path.join(d, "t")])
        out = subprocess.check_output([os.path.join(d, "t")]).split()
    got = [int(x) for x in out]
    assert got == [C.sizeof(_abi.Config), _abi.Config.seed.offset, _abi.Config.inbox_cap.offset,
                   _abi.Config.n_shards.offset, C.sizeof(_abi.Event), C.sizeof(_abi.Member)]


def test_no_gpu_means_loud_failure_not_fallback():
    """On a box without a HIP device swimsim_create must fail with SWIMSIM_ERR_DEVICE."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from swim_amd import SimConfig, configure
    err, sim = configure(SimConfig(nMembers=128))
    assert sim is None and err
    assert "no HIP device" in err or "hip" in err.lower()


def test_product_package_does_not_reference_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "swim_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "swimoracle_" not in text.replace('"swimoracle_"', "").replace("``swimoracle_``", ""), f
                assert "libswim_oracle" not in text, f
