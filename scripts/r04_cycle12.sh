#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
TAG=r04m
timeout 600 python scripts/graph_time.py 2>&1 | grep -v amdgpu.ids | tee $O/${TAG}_graph_time.txt
TICKS=100 timeout 300 python scripts/section_clocks.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_section_clocks_saturated.json; tail -c 3000 $O/${TAG}_section_clocks_saturated.json
