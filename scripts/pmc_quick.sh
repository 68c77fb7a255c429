#!/bin/bash
# one rocprofv3 --pmc pass per counter group over scripts/quick_time.py for each library given; summary of the tick kernels
# usage: pmc_quick.sh <tag> "<counters group 1>" "<counters group 2>" -- lib1.so lib2.so ...   (env WARM TICKS LOSS GC)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; TAG=$1; shift
groups=(); while [[ "$1" != "--" ]]; do groups+=("$1"); shift; done; shift
cd /tmp; export TMPDIR=/tmp WARM=${WARM:-200} TICKS=${TICKS:-40}
rm -rf /tmp/pq_$TAG; mkdir -p /tmp/pq_$TAG
for lib in "$@"; do
  g=0
  for grp in "${groups[@]}"; do
    g=$((g+1)); d=/tmp/pq_$TAG/$(basename $lib .so)_g$g
    timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $d -o p -- python $R/scripts/quick_time.py $R/swim_amd/csrc/$lib > $d.log 2>&1
  done
done
python $R/scripts/pmc_summary.py /tmp/pq_$TAG 30 2>&1 | grep -v "begin_kernel\|digest_kernel" > $O/${TAG}_pmc.txt; cat $O/${TAG}_pmc.txt
