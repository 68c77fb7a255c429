#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
TAG=r04b
timeout 600 python scripts/bounded_time.py 262144 64 2097152 64 2097152 128 2097152 256 2>&1 | grep -v amdgpu.ids | tee $O/${TAG}_bounded_time.txt
timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 2>&1 | tail -20 | tee $O/${TAG}_pytest_gpu.log
