#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
TAG=r04o
timeout 1200 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "strict" 2>&1 | tail -6 | tee $O/${TAG}_pytest_strict.log
(echo "# strict_reference_rules, 1 M members, saturated regime (every delivery an explicit record, todo lists applied in canonical order):"; STRICT=1 TICKS=50 timeout 600 python scripts/quick_time.py; echo "# the default (merge) on the same cluster:"; TICKS=50 timeout 300 python scripts/quick_time.py; echo "# 1 % loss, strict / default:"; STRICT=1 LOSS=10000 TICKS=30 timeout 600 python scripts/quick_time.py; LOSS=10000 TICKS=30 timeout 600 python scripts/quick_time.py) 2>&1 | grep -v amdgpu.ids | tee $O/${TAG}_strict_time.txt
timeout 300 scripts/microbench/random_access 30 8 2>&1 | tee $O/${TAG}_microbench_random_access.txt
