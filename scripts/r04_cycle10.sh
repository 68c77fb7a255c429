#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
TAG=r04k
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "bounded" 2>&1 | tail -5 | tee $O/${TAG}_pytest_bounded.log
(echo "# swimsim_cluster_step (exchange on the handles' streams):"; MEMBERS=4194304 SHARDS=4 timeout 600 python scripts/config5_cluster_one_gpu.py; timeout 900 python scripts/config5_cluster_one_gpu.py;
 echo "# phase calls + LocalFabric (host in the loop: SWIMSIM_CLUSTER_STEP=0):"; SWIMSIM_CLUSTER_STEP=0 timeout 900 python scripts/config5_cluster_one_gpu.py) 2>&1 | grep -v amdgpu.ids | tee $O/${TAG}_config5_cluster_one_gpu.txt
