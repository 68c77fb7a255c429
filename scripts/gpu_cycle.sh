#!/bin/bash
# one GPU iteration: parity tests, bench lines (both regimes), kernel-trace stats.
# usage: gpu_cycle.sh <tag> [pytest-args]   -> everything lands in gpurun_out/<tag>_*
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q "$@" 2>&1 | tail -8 | tee $O/${TAG}_pytest.log
timeout 300 python bench.py --steps 300 --warmup 150 > $O/${TAG}_bench_saturated.json 2> $O/${TAG}_bench_saturated.err; tail -c 1500 $O/${TAG}_bench_saturated.json
timeout 200 python bench.py --steps 300 --warmup 150 --regime quiescent --no-cpu-baseline > $O/${TAG}_bench_quiescent.json 2> $O/${TAG}_bench_quiescent.err; tail -c 600 $O/${TAG}_bench_quiescent.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof -o sat -- python $R/bench.py --steps 100 --warmup 150 --no-cpu-baseline > $O/${TAG}_prof.log 2>&1
cut -d, -f1-4,6-7 $O/${TAG}_prof/sat_kernel_stats.csv 2>/dev/null | head -6
