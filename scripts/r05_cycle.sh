#!/bin/bash
# round-5 GPU cycles (one gpurun call each): r05_cycle.sh <name>; output under gpurun_out/r05<name>_*
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
C=swim_amd/csrc
case "$1" in
a)  # view-cell layout A/B (SWIM_VSPLIT), random-sector rate by table size, BASELINE.md's config 3(s) as written
  (cd scripts/microbench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o random_access random_access.hip && for lg in 24 27 28 30; do ./random_access $lg; done) 2>&1 | tee $O/r05a_microbench_random_access_by_size.txt
  (ROUNDS=7 timeout 600 python scripts/ab_time.py $C/libswimsim.so $C/libswimsim_x_vsplit.so; echo '# 1 % loss:'; LOSS=10000 ROUNDS=5 CHUNK=20 timeout 600 python scripts/ab_time.py $C/libswimsim.so $C/libswimsim_x_vsplit.so) 2>&1 | grep -v amdgpu.ids | tee $O/r05a_ab_vsplit.txt
  timeout 900 python bench.py --crashes-per-tick 9.5 --gc --max-subjects 8192 --steps 300 --warmup 100 2>&1 | grep -v amdgpu.ids | tee $O/r05a_bench_config3s_as_written.json
  timeout 300 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "golden or config1 or small_populations" 2>&1 | tail -3 | tee $O/r05a_pytest_subset.log
  ;;
b)  # the new dense exchange: sharded GPU tests, one population as 1 / 2 / 4 / 8 handles, config 4 at full size
  timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "sharded or shard or golden" 2>&1 | tail -5 | tee $O/r05b_pytest_sharded.log
  KERNELS=1 timeout 900 python scripts/shard_time.py 1 2 4 8 2>&1 | grep -v amdgpu.ids | tee $O/r05b_shard_overhead_one_gpu.txt
  timeout 1200 python scripts/config4_one_gpu.py 2>&1 | grep -v amdgpu.ids | tee $O/r05b_config4_one_gpu.txt
  ;;
c)  # dense shards after the begin_kernel fix: overhead by handle count, where a sharded tick goes (kernel trace), bench.py --gpus 2 in one process
  KERNELS=1 timeout 900 python scripts/shard_time.py 1 2 4 8 2>&1 | grep -v amdgpu.ids | tee $O/r05c_shard_overhead_one_gpu.txt
  (cd /tmp && export TMPDIR=/tmp && for G in 2 8; do FORMS=cluster WARM=100 TICKS=30 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c$G -o t -- python $R/scripts/shard_time.py $G > /dev/null 2>&1; echo "# $G handles of one 1 048 576-member population, swimsim_cluster_step (130 ticks; Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs):"; cut -d, -f1-7 $(find /tmp/prof_c$G -name "*kernel_stats.csv" | head -1) | head -14; done) 2>&1 | tee $O/r05c_rocprof_sharded_kernels.txt
  SWIM_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 2 --steps 20 --warmup 5 --members 524288 2>&1 | grep -v amdgpu.ids | tee $O/r05c_bench_gpus2_one_process_shared_gpu.json
  ;;
esac
