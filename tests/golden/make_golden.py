"""Generates tests/golden/*.json from the CPU oracle (the reference itself cannot be run: no GHC
in the image).  These fixtures are a regression pin for the oracle and the common
target the GPU tests compare against.  Usage: python tests/golden/make_golden.py"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from tests import oracle_binding  # noqa: E402
from tests.test_oracle_semantics import run_fixture  # noqa: E402

SPECS = {
    # BASELINE config 1: 128 members, k=3 indirect probes, 1 injected failure
    "config1_n128_k3": {"n": 128, "k": 3, "seed": 1, "loss_ppm": 0, "suspicion": 21, "ticks": 200, "every": 10,
                        "faults": [[10, 64, 0]]},
    "lossy_n96_k3": {"n": 96, "k": 3, "seed": 7, "loss_ppm": 200000, "suspicion": 8, "ticks": 120, "every": 10,
                     "faults": [[4, 10, 0], [9, 50, 0]]},
    "churn_n64_k2": {"n": 64, "k": 2, "seed": 9, "loss_ppm": 50000, "suspicion": 6, "ticks": 150, "every": 10,
                     "faults": [[3, 5, 0], [30, 5, 1], [60, 5, 0], [90, 5, 1], [20, 33, 0]]},
    # the robust (round-robin) target scheme, with loss, a crash and a rejoin
    "robust_n96_k3": {"n": 96, "k": 3, "seed": 11, "loss_ppm": 100000, "suspicion": 7, "ticks": 120, "every": 10, "scheme": 1,
                      "faults": [[4, 10, 0], [9, 50, 0], [60, 10, 1]]},
    # bounded member maps (view_cap = 8) under 30 % loss: evictions every tick, a crash and a rejoin
    "bounded_n96_cap8": {"n": 96, "k": 3, "seed": 13, "loss_ppm": 300000, "suspicion": 6, "ticks": 60, "every": 5, "view_cap": 8,
                         "faults": [[4, 10, 0], [9, 50, 0], [30, 10, 1]]},
    # strict_reference_rules (the literal suspectOrDeadNode' under the canonical order) where the rules part: 20 % loss, a rejoin
    "strict_n96_k3": {"n": 96, "k": 3, "seed": 17, "loss_ppm": 200000, "suspicion": 6, "ticks": 100, "every": 10, "strict": 1,
                      "faults": [[4, 10, 0], [9, 50, 0], [40, 10, 1]]},
    # a push-pull every 5 periods (the periodic state exchange, both halves)
    "pushpull_n96_k3": {"n": 96, "k": 3, "seed": 19, "loss_ppm": 100000, "suspicion": 6, "ticks": 100, "every": 10, "pull_ticks": 5, "push_pull": 1,
                        "faults": [[4, 10, 0], [9, 50, 0], [40, 10, 1]]},
}

if __name__ == "__main__":
    abi = oracle_binding.load()
    for name, spec in SPECS.items():
        out = {"generator": "tests/golden/make_golden.py (oracle/swim_oracle.c)", "spec": spec,
               "expect": run_fixture(abi, spec)}
        with open(os.path.join(HERE, name + ".json"), "w") as f:
            json.dump(out, f, indent=1, sort_keys=True)
        print(name, out["expect"]["digests"][-1], out["expect"]["n_events"])
