#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
TAG=r04j
C=swim_amd/csrc
(echo "# merge_kernel: members dealt to threads by work (SWIM_MERGE_SORT=1) against the plain assignment; saturated, 1 M members"; ROUNDS=7 timeout 400 python scripts/ab_time.py $C/libswimsim.so $C/libswimsim_x_msort.so;
 echo "# 1 % loss + settling"; LOSS=10000 GC=1 WARM=100 CHUNK=20 ROUNDS=5 timeout 400 python scripts/ab_time.py $C/libswimsim.so $C/libswimsim_x_msort.so;
 echo "# 2 M members"; MEMBERS=2097152 WARM=120 CHUNK=30 ROUNDS=5 timeout 400 python scripts/ab_time.py $C/libswimsim.so $C/libswimsim_x_msort.so) 2>&1 | grep -v amdgpu.ids | tee $O/${TAG}_ab_merge_sort.txt
