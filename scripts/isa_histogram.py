"""Static instruction mix of the gfx950 kernels (no GPU needed): hipcc -S of swimsim.hip, instructions of each kernel
by class (environment DEFS="-DSWIM_STATE_BY_POINTER ..." compiles a variant).  Static counts, not executed ones -- a map of what the code is made of (quarter-rate integer multiplies of the
hashes, lane moves of spilled scalars, waits) to read next to the section clocks.  usage: isa_histogram.py [kernel ...]"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = sys.argv[1:] or ["merge_kernel", "probe_kernelILi4", "publish_kernel", "xlat_kernel", "ingest_kernel"]
CLASSES = [("int mul (quarter rate)", ("v_mul_lo", "v_mul_hi", "v_mad_u64", "v_mad_i64")),
           ("lane moves (scalar spills, broadcasts)", ("v_readlane", "v_writelane", "v_readfirstlane")),
           ("s_waitcnt", ("s_waitcnt",)), ("global loads", ("global_load", "flat_load", "buffer_load")),
           ("global stores", ("global_store", "flat_store", "buffer_store")),
           ("global atomics", ("global_atomic", "flat_atomic", "buffer_atomic")), ("scratch (vector spills)", ("scratch_",)),
           ("LDS", ("ds_",)), ("DPP / cross-lane", ("v_mov_b32_dpp", "v_add_u32_dpp", "v_permlane")),
           ("branches", ("s_cbranch", "s_branch")), ("other scalar", ("s_",)), ("other vector", ("v_",))]

with tempfile.TemporaryDirectory() as tmp:
    asm = os.path.join(tmp, "swimsim.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", *os.environ.get("DEFS", "").split(),
                           "--cuda-device-only", "-S", "-o", asm,
                           os.path.join(ROOT, "swim_amd", "csrc", "swimsim.hip")], stderr=subprocess.DEVNULL, cwd=tmp)
    txt = open(asm).read()
parts = re.split(r"\n(_ZN4swim[^\n:]+):[^\n]*\n", txt)
for k in range(1, len(parts), 2):
    name, body = parts[k], parts[k + 1].split(".Lfunc_end")[0]
    if not any(x in name for x in KERNELS):
        continue
    ops, n = collections.Counter(), 0
    for line in body.split("\n"):
        line = line.strip()
        if not line or line[0] in ".;/" or line.endswith(":"):
            continue
        op = line.split()[0]
        n += 1
        ops[next((c for c, pre in CLASSES if op.startswith(pre)), "other")] += 1
    print("%s: %d static instructions" % (name, n))
    for c, v in ops.most_common():
        print("    %-42s %6d  %4.1f %%" % (c, v, 100.0 * v / n))
