#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
TAG=r04g
(for lib in swim_amd/csrc/libswimsim.so swim_amd/csrc/libswimsim_x_nospec.so; do for g in "2560,2560" "8192,8192" "16384,16384" "65536,65536" "262144,262144"; do echo "# $lib SWIMSIM_SP_GRID=$g"; LIB=$lib SWIMSIM_SP_GRID=$g TICKS=10 timeout 120 python scripts/bounded_time.py 2097152 64; done; done) 2>&1 | grep -v amdgpu.ids | tee $O/${TAG}_bounded_grid_sweep.txt
