#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
TAG=r04fin4
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "sharded_cluster_of_bounded or config1 or golden or strict_reference_rules_on_the_gpu and 777" 2>&1 | tail -3 | tee $O/${TAG}_pytest_some.log
cd /tmp && export TMPDIR=/tmp
PASSES="p1 p2 p4" bash $R/scripts/pmc_passes.sh $O/${TAG}_pmc > $O/${TAG}_pmc.log 2>&1; tail -12 $O/${TAG}_pmc/summary.txt | cut -c1-200
bash $R/scripts/prof_timed_window.sh $TAG --steps 20 --warmup 5 | tail -12
