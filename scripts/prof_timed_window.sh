#!/bin/bash
# rocprofv3 --kernel-trace of bench.py, statistics over the TIMED WINDOW only (the last `steps` ticks before the
# digest): per-kernel mean duration and the gaps between the tick's kernels -- what `roofline.kernels.*.avg_launch_us`
# must reproduce (the pre-roll and warm-up dispatches are dropped).  usage: prof_timed_window.sh <tag> [bench args]
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; TAG=$1; shift
STEPS=20; W=5; args=("$@"); [[ ${#args[@]} -eq 0 ]] && args=(--steps $STEPS --warmup $W)
for ((k=0;k<${#args[@]};k++)); do [[ "${args[$k]}" == "--steps" ]] && STEPS=${args[$((k+1))]}; done
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/ptw_$TAG
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ptw_$TAG -o t -- python $R/bench.py "${args[@]}" --no-cpu-baseline > $O/${TAG}_rocprof_bench_line.json 2> /tmp/ptw_$TAG.err
f=$(find /tmp/ptw_$TAG -name "*kernel_trace.csv" | head -1)
{ echo "# rocprofv3 --kernel-trace -- python bench.py ${args[*]} --no-cpu-baseline ; statistics over the last $STEPS ticks (the timed window)";
  python $R/scripts/trace_gaps.py $f $STEPS;
  echo "# whole-run --stats (pre-roll and warm-up included), for comparison:";
  cut -d, -f1-4,6-7 $(find /tmp/ptw_$TAG -name "*kernel_stats.csv" | head -1) | head -6;
  echo "# the bench line of the same run (HIP events inside the library):";
  python -c "import json,sys; o=json.load(open('$O/${TAG}_rocprof_bench_line.json')); print({k: round(v['avg_launch_us'],1) for k,v in o['roofline']['kernels'].items()}, 'ms_per_step', round(o['ms_per_step'],4))"; } | tee $O/${TAG}_rocprof_timed_window.txt
