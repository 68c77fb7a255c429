"""Exchange records per tick of a sharded cluster, record path against replicated queue masks (CPU only: the
product's kernels in the host emulation, counts are exact).  usage: shard_record_counts.py [members] [shards]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(rm, n, shards):
    os.environ["SWIMSIM_SHARD_REPLICATED_MASKS"] = rm
    from swim_amd import Config, SimConfig, workloads
    from swim_amd.shard import LocalFabric, ShardedSim, REC_BYTES
    from tests import hostemu_binding
    crashes = workloads.hashed_crashes(n, 5, 1, 40, 3, 200)
    sc = SimConfig(cfg=Config(numToGossip=3), nMembers=n, seed=5, eventMask=0, suspicionTicks=7, maxSubjects=1024)
    tot = {}

    class Counting(LocalFabric):
        def exchange(self, shards_, kinds, counts):
            for si in range(len(shards_)):
                for j, k in enumerate(kinds):
                    tot[k] = tot.get(k, 0) + sum(counts[si][j])
            return super().exchange(shards_, kinds, counts)
    b = ShardedSim(hostemu_binding.load(), sc, Counting(shards))
    workloads.apply_crashes(b, crashes)
    b.step(40)
    tot.clear()
    c0 = b.counters()
    b.step(20)
    c1 = b.counters()
    per_tick = {k: v / 20 for k, v in sorted(tot.items())}
    print("replicated_masks=%s members=%d shards=%d  records/tick by kind %s  bytes/tick %d  payloads/member-tick %.3f  digest %016x" % (
        rm, n, shards, per_tick, sum(v * REC_BYTES[k] for k, v in per_tick.items()),
        (c1["payloads"] - c0["payloads"]) / 20 / n, b.digest()))


if __name__ == "__main__":
    if len(sys.argv) > 3:
        run(sys.argv[3], int(sys.argv[1]), int(sys.argv[2]))
    else:
        n, shards = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (8192, 4)
        for rm in ("0", "1"):                         # the knob is read at create: one process per setting
            subprocess.check_call([sys.executable, os.path.abspath(__file__), str(n), str(shards), rm])
