#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
TAG=r04fin6
timeout 1500 python -m pytest tests -m gpu -x -q --durations=5 2>&1 | tail -10 | tee $O/${TAG}_pytest.log
cd /tmp && export TMPDIR=/tmp
PASSES="p1 p2 p4" bash $R/scripts/pmc_passes.sh $O/${TAG}_pmc > $O/${TAG}_pmc.log 2>&1; tail -12 $O/${TAG}_pmc/summary.txt | cut -c1-160
bash $R/scripts/prof_timed_window.sh $TAG --steps 20 --warmup 5 | tail -3
