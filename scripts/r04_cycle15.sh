#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
TAG=r04q
(echo "# sp_probe_lane_kernel (one member per lane; default):"; timeout 600 python scripts/bounded_time.py 65536 64 1048576 64 2097152 64 2097152 128 2097152 256 4194304 64; echo "# sp_probe_kernel (one wave per member; SWIMSIM_SP_PROBE=wave):"; SWIMSIM_SP_PROBE=wave timeout 600 python scripts/bounded_time.py 2097152 64 2097152 256; echo "# numToGossip 10, 262144 members:"; K=10 timeout 600 python scripts/bounded_time.py 262144 64; K=10 SWIMSIM_SP_PROBE=wave timeout 600 python scripts/bounded_time.py 262144 64; echo "# lossless:"; LOSS=0 timeout 600 python scripts/bounded_time.py 2097152 64;  LOSS=0 SWIMSIM_SP_PROBE=wave timeout 600 python scripts/bounded_time.py 2097152 64) 2>&1 | grep -v amdgpu.ids | tee $O/${TAG}_bounded_time_lane_vs_wave.txt
timeout 1500 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "bounded" 2>&1 | tail -5 | tee $O/${TAG}_pytest_bounded.log
