#!/bin/bash
# rocprofv3 --kernel-trace of bench.py, statistics over the TIMED WINDOW only (the last `steps` ticks before the
# digest): per-kernel mean duration and the gaps between the tick's kernels -- what `roofline.kernels.*.avg_launch_us`
# must reproduce (the pre-roll and warm-up dispatches are dropped).  usage: prof_timed_window.sh <tag> [bench args]
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; TAG=$1; shift
STEPS=20; W=5; args=("$@"); [[ ${#args[@]} -eq 0 ]] && args=(--steps $STEPS --warmup $W)
for ((k=0;k<${#args[@]};k++)); do [[ "${args[$k]}" == "--steps" ]] && STEPS=${args[$((k+1))]}; done
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/ptw_$TAG
# (the default run of bench.py measures TWO clusters since round 6: the headline is traced with --no-as-written, so that "the last $STEPS
# ticks" are its timed window; AS_WRITTEN=1 traces the default run instead, whose last ticks are BASELINE.md row 3(s) as written)
EXTRA="--no-as-written"; [[ -n "$AS_WRITTEN" ]] && EXTRA=""
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ptw_$TAG -o t -- python $R/bench.py "${args[@]}" --no-cpu-baseline $EXTRA > $O/${TAG}_rocprof_bench_line.json 2> /tmp/ptw_$TAG.err
f=$(find /tmp/ptw_$TAG -name "*kernel_trace.csv" | head -1)
{ echo "# rocprofv3 --kernel-trace -- python bench.py ${args[*]} --no-cpu-baseline $EXTRA ; statistics over the last $STEPS ticks (the timed window${AS_WRITTEN:+ of config3s_as_written})";
  python $R/scripts/trace_gaps.py $f $STEPS;
  echo "# whole-run --stats (pre-roll and warm-up included), for comparison:";
  cut -d, -f1-4,6-7 $(find /tmp/ptw_$TAG -name "*kernel_stats.csv" | head -1) | head -6;
  echo "# the bench line of the same run (HIP events inside the library):";
  python -c "import json,sys; o=json.load(open('$O/${TAG}_rocprof_bench_line.json')); o=o.get('config3s_as_written', o) if '$AS_WRITTEN' else o; print({k: round(v['avg_launch_us'],1) for k,v in o['roofline']['kernels'].items()}, 'ms_per_step', round(o['ms_per_step'],4))"; } | tee $O/${TAG}_rocprof_timed_window.txt
