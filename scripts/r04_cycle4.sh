#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
TAG=r04d
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "bounded or config5_with" 2>&1 | tail -5 | tee $O/${TAG}_pytest_bounded.log
timeout 600 python scripts/bounded_time.py 262144 64 2097152 64 2097152 128 2097152 256 2>&1 | grep -v amdgpu.ids | tee $O/${TAG}_bounded_time.txt
timeout 300 python scripts/bounded_sections.py 2097152 64 2>&1 | grep -v amdgpu.ids > $O/${TAG}_bounded_sections_2m_cap64.json
timeout 300 python scripts/bounded_sections.py 2097152 256 2>&1 | grep -v amdgpu.ids > $O/${TAG}_bounded_sections_2m_cap256.json
