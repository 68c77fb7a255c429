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
d)  # dense shards, ingest flattened: strong (1 M members as G handles) and weak (G x 1 M members vs ONE handle of G M) on one GPU; kernel stats; config 5 with timeouts below the entries' lifetime
  (echo "# strong: 1 048 576 members as G handles"; FORMS=cluster KERNELS=1 timeout 900 python scripts/shard_time.py 1 2 4 8;
   for G in 2 4 8; do echo "# weak: $G x 1 048 576 members vs one handle of $((G * 1048576))"; MEMBERS=$((G * 1048576)) FORMS=cluster WARM=100 TICKS=30 timeout 900 python scripts/shard_time.py 1 $G; done) 2>&1 | grep -v amdgpu.ids | tee $O/r05d_shard_overhead_one_gpu.txt
  (cd /tmp && export TMPDIR=/tmp && for G in 2 8; do FORMS=cluster WARM=100 TICKS=30 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_d$G -o t -- python $R/scripts/shard_time.py $G > /dev/null 2>&1; echo "# $G handles of one 1 048 576-member population, swimsim_cluster_step (130 ticks), rocprofv3 --kernel-trace --stats:"; python $R/scripts/kernel_stats.py /tmp/prof_d$G 10; done) 2>&1 | tee $O/r05d_rocprof_sharded_kernels.txt
  (for cs in "256 8" "256 16" "128 6"; do set -- $cs; echo "# view_cap $1, suspicion timeout $2 ticks, 262 144 members, oracle-checked:"; CAP=$1 S=$2 TICKS=80 T0=30 ORACLE=1 CHURN=0,10 timeout 1200 python scripts/config5.py 262144; done) 2>&1 | grep -v amdgpu.ids | tee $O/r05d_config5_bounded_short_timeouts.txt
  ;;
e)  # sharded probe with batched replica gathers; grid barrier vs kernel boundary; the whole GPU suite
  (cd scripts/microbench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o grid_barrier grid_barrier.hip 2>/dev/null && timeout 120 ./grid_barrier 20 50) 2>&1 | tee $O/r05e_microbench_grid_barrier.txt
  (echo "# strong: 1 048 576 members as G handles"; FORMS=cluster KERNELS=1 timeout 900 python scripts/shard_time.py 1 2 4 8;
   for G in 2 4; do echo "# weak: $G x 1 048 576 members vs one handle of $((G * 1048576))"; MEMBERS=$((G * 1048576)) FORMS=cluster WARM=100 TICKS=30 timeout 900 python scripts/shard_time.py 1 $G; done) 2>&1 | grep -v amdgpu.ids | tee $O/r05e_shard_overhead_one_gpu.txt
  (cd /tmp && export TMPDIR=/tmp && for G in 2; do FORMS=cluster WARM=100 TICKS=30 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_e$G -o t -- python $R/scripts/shard_time.py $G > /dev/null 2>&1; echo "# $G handles of one 1 048 576-member population, swimsim_cluster_step (130 ticks), rocprofv3 --kernel-trace --stats:"; python $R/scripts/kernel_stats.py /tmp/prof_e$G 8; done) 2>&1 | tee $O/r05e_rocprof_sharded_kernels.txt
  timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $O/r05e_pytest_gpu.log
  ;;
f)  # sp_merge_kernel with the next member's inputs prefetched (A/B against the library of cycle e); sharded probe with register-held Ping records
  (for L in $C/libswimsim_x_before_prefetch.so $C/libswimsim.so; do echo "# $L"; LIB=$L timeout 600 python scripts/bounded_time.py 2097152 64 1048576 256 2097152 128; done) 2>&1 | grep -v amdgpu.ids | tee $O/r05f_bounded_time_prefetch_ab.txt
  (echo "# strong: 1 048 576 members as G handles"; FORMS=cluster KERNELS=1 timeout 900 python scripts/shard_time.py 1 2 4) 2>&1 | grep -v amdgpu.ids | tee $O/r05f_shard_overhead_one_gpu.txt
  (cd /tmp && export TMPDIR=/tmp && for G in 2; do FORMS=cluster WARM=100 TICKS=30 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_f$G -o t -- python $R/scripts/shard_time.py $G > /dev/null 2>&1; echo "# $G handles of one 1 048 576-member population, swimsim_cluster_step (130 ticks), rocprofv3 --kernel-trace --stats:"; python $R/scripts/kernel_stats.py /tmp/prof_f$G 8; done) 2>&1 | tee $O/r05f_rocprof_sharded_kernels.txt
  ;;
g)  # sp_merge_kernel with its wave-uniform values in scalar registers (A/B against the library of cycle e)
  (for L in $C/libswimsim_x_r05e.so $C/libswimsim.so; do echo "# $L"; LIB=$L timeout 600 python scripts/bounded_time.py 2097152 64 1048576 256 2097152 128; done) 2>&1 | grep -v amdgpu.ids | tee $O/r05g_bounded_time_scalarized_ab.txt
  ;;
h)  # where a shard's tick goes at the per-GPU size: 8 x 1 048 576 members and 4 x 1 048 576 on one GPU, kernel statistics
  (cd /tmp && export TMPDIR=/tmp && for G in 4 8; do MEMBERS=$((G * 1048576)) FORMS=cluster WARM=60 TICKS=20 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_h$G -o t -- python $R/scripts/shard_time.py $G > /tmp/h$G.out 2>&1; grep shards_on /tmp/h$G.out; echo "# $G handles x 1 048 576 members, swimsim_cluster_step (80 ticks), rocprofv3 --kernel-trace --stats:"; python $R/scripts/kernel_stats.py /tmp/prof_h$G 8; done) 2>&1 | tee $O/r05h_rocprof_sharded_kernels_weak.txt
  ;;
fin)  # the cycle of the FINAL kernels: tests, the bench lines, kernel trace over the timed window, PMC traffic, the multi-GPU forms on one GPU
  bash scripts/gpu_cycle.sh r05fin tests bench extra prof pmc
  SWIMSIM_FOLD_BEGIN=0 timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/r05fin_bench_driver_flags_with_begin_kernel.json 2>/dev/null
  SWIMSIM_FOLD_BEGIN=0 timeout 400 python bench.py --steps 300 --warmup 150 --no-cpu-baseline > $O/r05fin_bench_saturated_with_begin_kernel.json 2>/dev/null
  for G in 2 4; do SWIM_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus $G --steps 20 --warmup 5 2>/dev/null | grep -v amdgpu.ids > $O/r05fin_bench_gpus${G}_one_process_shared_gpu.json; done
  (echo "# strong: 1 048 576 members as G handles"; FORMS=cluster,phases KERNELS=1 timeout 900 python scripts/shard_time.py 1 2 4 8;
   for G in 2 4 8; do echo "# weak: $G x 1 048 576 members vs one handle of $((G * 1048576))"; MEMBERS=$((G * 1048576)) FORMS=cluster WARM=100 TICKS=30 timeout 900 python scripts/shard_time.py 1 $G; done) 2>&1 | grep -v amdgpu.ids | tee $O/r05fin_shard_overhead_one_gpu.txt
  timeout 1200 python scripts/config4_one_gpu.py 2>&1 | grep -v amdgpu.ids | tee $O/r05fin_config4_one_gpu.txt
  ;;
i)  # merge_kernel's stores as non-temporal stores (A/B on one cluster: kernels and wall)
  (ROUNDS=7 timeout 600 python scripts/ab_time.py $C/libswimsim_x_base.so $C/libswimsim_x_nt.so) 2>&1 | grep -v amdgpu.ids | tee $O/r05i_ab_nt_stores.txt
  ;;
fin2)  # tests, bench lines, kernel trace and PMC once more after the round's last source change
  bash scripts/gpu_cycle.sh r05fin2 tests bench prof pmc
  ;;
fin3)  # tests, bench lines, kernel trace and PMC on the round's LAST sources; the one-process multi-GPU line once more
  bash scripts/gpu_cycle.sh r05fin3 tests bench prof pmc
  SWIM_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 2 --steps 20 --warmup 5 2>/dev/null | grep -v amdgpu.ids > $O/r05fin3_bench_gpus2_one_process_shared_gpu.json
  (FORMS=cluster KERNELS=1 timeout 600 python scripts/shard_time.py 1 2) 2>&1 | grep -v amdgpu.ids | tee $O/r05fin3_shard_overhead_one_gpu.txt
  ;;
j)  # the sharded probe by sections against an unsharded handle of the same size
  timeout 600 python scripts/section_clocks_sharded.py 2 2>&1 | grep -v amdgpu.ids | tee $O/r05j_section_clocks_sharded_probe.txt
  ;;
k)  # the sharded probe's routing tail behind LDS-only barriers
  (echo "# strong: 1 048 576 members as G handles"; FORMS=cluster KERNELS=1 timeout 900 python scripts/shard_time.py 1 2 4 8;
   for G in 2 4; do echo "# weak: $G x 1 048 576 members vs one handle of $((G * 1048576))"; MEMBERS=$((G * 1048576)) FORMS=cluster WARM=100 TICKS=30 timeout 900 python scripts/shard_time.py 1 $G; done) 2>&1 | grep -v amdgpu.ids | tee $O/r05k_shard_overhead_one_gpu.txt
  (cd /tmp && export TMPDIR=/tmp && for G in 2; do FORMS=cluster WARM=100 TICKS=30 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_k$G -o t -- python $R/scripts/shard_time.py $G > /dev/null 2>&1; echo "# $G handles of one 1 048 576-member population, swimsim_cluster_step (130 ticks), rocprofv3 --kernel-trace --stats:"; python $R/scripts/kernel_stats.py /tmp/prof_k$G 8; done) 2>&1 | tee $O/r05k_rocprof_sharded_kernels.txt
  ;;
l)  # the known-ring learns the rumours a member states itself (A/B on one cluster)
  (ROUNDS=7 timeout 600 python scripts/ab_time.py $C/libswimsim_x_noown.so $C/libswimsim_x_own.so; echo '# 1 % loss:'; LOSS=10000 ROUNDS=5 CHUNK=20 timeout 600 python scripts/ab_time.py $C/libswimsim_x_noown.so $C/libswimsim_x_own.so) 2>&1 | grep -v amdgpu.ids | tee $O/r05l_ab_own_known.txt
  ;;
fin4)  # the round's last kernels (the known-ring learns what a member states itself): tests, bench lines, trace, PMC, the sharded forms
  bash scripts/gpu_cycle.sh r05fin4 tests bench prof pmc
  SWIM_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 2 --steps 20 --warmup 5 2>/dev/null | grep -v amdgpu.ids > $O/r05fin4_bench_gpus2_one_process_shared_gpu.json
  (FORMS=cluster KERNELS=1 timeout 600 python scripts/shard_time.py 1 2 4) 2>&1 | grep -v amdgpu.ids | tee $O/r05fin4_shard_overhead_one_gpu.txt
  ;;
esac
