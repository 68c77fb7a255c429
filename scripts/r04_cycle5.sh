#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
TAG=r04e
C=swim_amd/csrc
(echo "# P = 3 (headline): probe_kernel<4> at 5 waves (96 VGPRs + 12 B scratch) against 4 waves (99 VGPRs, none)"; ROUNDS=5 timeout 300 python scripts/ab_time.py $C/libswimsim.so $C/libswimsim_x_pw4.so;
 echo "# P = 10 (the reference's default): probe_kernel<12> chunked, 3 waves (162 VGPRs) against 4 waves (128 VGPRs + 68 B scratch)"; P=10 WARM=100 CHUNK=20 ROUNDS=5 timeout 400 python scripts/ab_time.py $C/libswimsim.so $C/libswimsim_x_p12w4.so) 2>&1 | grep -v amdgpu.ids | tee $O/${TAG}_ab_probe_waves.txt
timeout 300 python bench.py --steps 20 --warmup 5 --members 2097152 --loss-ppm 300000 --view-cap 64 > $O/${TAG}_bench_config5_cap64.json 2> $O/${TAG}_bench_config5_cap64.err; tail -c 1500 $O/${TAG}_bench_config5_cap64.json; tail -2 $O/${TAG}_bench_config5_cap64.err
timeout 600 python scripts/bounded_time.py 2097152 64 2097152 256 2>&1 | grep -v amdgpu.ids | tee $O/${TAG}_bounded_time.txt
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "full_event_stream or small_populations or config1" 2>&1 | tail -5 | tee $O/${TAG}_pytest_some.log
