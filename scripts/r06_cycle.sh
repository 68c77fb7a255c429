#!/bin/bash
# round-6 GPU cycles (one gpurun call each): r06_cycle.sh <name>; output under gpurun_out/r06<name>_*
# Every cycle starts with the smoke run (gpu_cycle.sh's canary, or the one below): nothing else is started on a box that fails it.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
C=swim_amd/csrc
canary() { if ! timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/r06$1_canary.log 2>&1; then tail -5 $O/r06$1_canary.log; echo "r06_cycle: the smoke run failed on this box -- nothing else is started"; exit 3; fi; ulimit -c 0; }
case "$1" in
a)  # the two patches of round 5 (a message from outside opens its row on every shard; the known-ring learns what a member states): suite, bench lines, trace, PMC
  bash scripts/gpu_cycle.sh r06a tests bench prof pmc
  ;;
esac
