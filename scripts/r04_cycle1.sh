#!/bin/bash
# round 4, first GPU cycle: the new bounded-map tests first (fail fast), then the whole GPU suite, the bench line, config 5 at 2 M members
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
TAG=r04a
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "bounded or config5_with" --durations=8 2>&1 | tail -25 | tee $O/${TAG}_pytest_bounded.log
timeout 300 python bench.py --steps 20 --warmup 5 > $O/${TAG}_bench_driver_flags.json 2> $O/${TAG}_bench_driver_flags.err; tail -c 2500 $O/${TAG}_bench_driver_flags.json; tail -2 $O/${TAG}_bench_driver_flags.err
(CAP=64 TICKS=160 T0=60 timeout 600 python scripts/config5.py 2097152; CAP=256 TICKS=160 T0=60 CHURN=0 timeout 600 python scripts/config5.py 2097152) 2>&1 | grep -v amdgpu.ids | tee $O/${TAG}_config5_bounded_2m.txt
timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 2>&1 | tail -20 | tee $O/${TAG}_pytest_gpu.log
