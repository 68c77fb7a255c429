#!/bin/bash
# one GPU iteration: parity tests, wall-clock probe, kernel-trace stats.  usage: gpu_cycle.sh <tag> [pytest-args]
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}; mkdir -p $R/gpurun_out
python -m pytest tests -m gpu -x -q "$@" 2>&1 | tail -6 | tee $R/gpurun_out/pytest_$TAG.log
python scripts/quick_time.py 0 1 2>&1 | tee $R/gpurun_out/quick_$TAG.log
cd /tmp && export TMPDIR=/tmp WARM=150 TICKS=100
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o loaded -- python $R/scripts/quick_time.py 1 > $R/gpurun_out/prof_$TAG.log 2>&1
cut -d, -f1-4,6-7 $R/gpurun_out/prof_$TAG/loaded_kernel_stats.csv | head -5
