"""The N > 1 path with one PROCESS per shard over torch.distributed (gloo, world_size 2 and 4, CPU):
the same host code (swim_amd/shard.py DistFabric) that runs RCCL on GPUs.  Kernels run through the
host emulation; rank 0 compares every observable with the oracle."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_world(world, args, port, **extra_env):
    from tests import hostemu_binding, oracle_binding
    hostemu_binding.build()
    oracle_binding.build()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1", **extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "dist_worker.py")] + [str(a) for a in args]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert "DIST-OK world=%d" % world in out.stdout, out.stdout[-2000:]


@pytest.mark.parametrize("world,n,p,loss,seed,port", [(2, 128, 3, 0, 1, 29611), (2, 256, 3, 100000, 2, 29612), (4, 256, 2, 50000, 3, 29613)])
def test_one_process_per_shard_gloo(world, n, p, loss, seed, port):
    run_world(world, (n, p, loss, seed, 45), port)


def test_one_process_per_shard_gloo_with_settling():
    """gc_ticks on: the third exchange round (settle records) through swimsim_shard_step's callback."""
    run_world(2, (192, 3, 20000, 7, 130, 1), 29614)


def test_one_process_per_shard_gloo_with_join_pull_and_settling():
    """join_pull (round 0: the hosts' entries travel to the joiners' owners ahead of the probes) + settling."""
    run_world(2, (192, 3, 20000, 8, 130, 3), 29615)


def test_one_process_per_shard_gloo_with_periodic_state_pulls():
    """pull_ticks on a sharded cluster: round 0 in EVERY tick (the periodic pullers whose hosts live on the other shard) through
    swimsim_shard_step's callback, with join pulls and settling on top."""
    run_world(2, (192, 3, 20000, 13, 60, 4), 29620)
    run_world(4, (256, 3, 100000, 14, 60, 7), 29621)


def test_one_process_per_shard_gloo_more_configurations():
    """The all-gather of dictionaries, queue masks and queue bytes (round 1) and the {dst, src} records (round 2) through
    swimsim_shard_step's callback, with loss, settling and the join-time pull on top."""
    run_world(4, (256, 3, 50000, 9, 60, 0), 29616)
    run_world(2, (192, 3, 20000, 8, 130, 3), 29617)


def test_one_process_per_shard_gloo_with_bounded_member_maps():
    """A cluster of bounded handles (view_cap = 16 / 64), one process per shard: the all-gather of queue lines
    and member bytes and the 8-byte delivery records through swimsim_shard_step's callback, 30 % loss, a crash and a rejoin."""
    run_world(2, (192, 3, 300000, 11, 40, 16 << 8), 29618)
    run_world(4, (256, 3, 300000, 12, 30, 64 << 8), 29619)


def test_one_process_per_shard_gloo_with_the_literal_rule():
    """strict_reference_rules on a sharded cluster, one process per shard: every queue travels as a list in round 1, every delivery is
    an explicit record; = the unsharded oracle's literal mode (15 % / 25 % loss: the rules part)."""
    run_world(2, (192, 3, 150000, 21, 60, 8), 29622)
    run_world(4, (256, 3, 250000, 22, 50, 8), 29623)


def test_one_process_per_shard_gloo_with_push_pull():
    """push_pull on a sharded cluster, one process per shard: the pullers' maps travel to their hosts' owners in round 0."""
    run_world(2, (192, 3, 20000, 15, 60, 4 | 16), 29624)
    run_world(4, (256, 3, 100000, 16, 60, 7 | 16), 29625)
