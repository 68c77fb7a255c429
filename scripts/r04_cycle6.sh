#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
TAG=r04f
(for g in "0,0" "1024,0" "1536,0" "1792,0" "2560,0" "4096,0" "8192,0" "0,768" "0,1024" "0,2048" "0,4096"; do echo "# SWIMSIM_SP_GRID=$g"; SWIMSIM_SP_GRID=$g TICKS=10 timeout 120 python scripts/bounded_time.py 2097152 64; done) 2>&1 | grep -v amdgpu.ids | tee $O/${TAG}_bounded_grid_sweep.txt
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "bounded or config5_with" 2>&1 | tail -4 | tee $O/${TAG}_pytest_bounded.log
