#!/bin/bash
# one GPU iteration: parity tests, bench lines, kernel-trace stats, PMC passes.
# usage: gpu_cycle.sh <tag> [what...]   what = tests bench prof pmc variants (default: all)   -> gpurun_out/<tag>_*
TAG=$1; shift; WHAT="${*:-tests bench prof pmc variants}"
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O
cd $R
has() { [[ " $WHAT " == *" $1 "* ]]; }
if has tests; then
  timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 2>&1 | tail -25 | tee $O/${TAG}_pytest.log
fi
if has bench; then
  timeout 300 python bench.py --steps 20 --warmup 5 > $O/${TAG}_bench_driver_flags.json 2> $O/${TAG}_bench_driver_flags.err; tail -c 2500 $O/${TAG}_bench_driver_flags.json; tail -3 $O/${TAG}_bench_driver_flags.err
  timeout 400 python bench.py --steps 300 --warmup 150 > $O/${TAG}_bench_saturated.json 2> $O/${TAG}_bench_saturated.err; tail -c 2500 $O/${TAG}_bench_saturated.json; tail -3 $O/${TAG}_bench_saturated.err
  timeout 200 python bench.py --steps 300 --warmup 150 --regime quiescent --no-cpu-baseline > $O/${TAG}_bench_quiescent.json 2> $O/${TAG}_bench_quiescent.err; tail -c 700 $O/${TAG}_bench_quiescent.json
  timeout 300 python bench.py --steps 300 --warmup 150 --gc --no-cpu-baseline > $O/${TAG}_bench_saturated_gc.json 2> $O/${TAG}_bench_saturated_gc.err; tail -c 900 $O/${TAG}_bench_saturated_gc.json; tail -3 $O/${TAG}_bench_saturated_gc.err
  timeout 300 python bench.py --steps 300 --warmup 150 --scheme robust --no-cpu-baseline > $O/${TAG}_bench_saturated_robust.json 2> $O/${TAG}_bench_saturated_robust.err; tail -c 700 $O/${TAG}_bench_saturated_robust.json
fi
if has variants; then
  WARM=150 TICKS=200 timeout 400 python scripts/quick_time.py $(ls swim_amd/csrc/libswimsim.so swim_amd/csrc/libswimsim_mw*.so swim_amd/csrc/libswimsim_pw*.so 2>/dev/null) 2>&1 | tee $O/${TAG}_variants.txt
fi
cd /tmp && export TMPDIR=/tmp
if has prof; then
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof -o sat -- python $R/bench.py --steps 100 --warmup 50 --no-cpu-baseline > $O/${TAG}_prof.log 2>&1
  cut -d, -f1-4,6-7 $O/${TAG}_prof/*/sat_kernel_stats.csv $O/${TAG}_prof/sat_kernel_stats.csv 2>/dev/null | head -8
fi
if has pmc; then
  bash $R/scripts/pmc_passes.sh $O/${TAG}_pmc > $O/${TAG}_pmc.log 2>&1; tail -40 $O/${TAG}_pmc/summary.txt
fi
