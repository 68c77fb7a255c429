#!/bin/bash
# one GPU iteration: parity tests, bench lines, kernel-trace stats, PMC passes.
# usage: gpu_cycle.sh <tag> [what...]   what = tests bench extra variants shard prof pmc churn config5 bounded cluster strict (default: all but churn, config5, bounded, cluster, strict)   -> gpurun_out/<tag>_*
TAG=$1; shift; WHAT="${*:-tests bench extra variants shard prof pmc}"
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O
cd $R
has() { [[ " $WHAT " == *" $1 "* ]]; }
# canary (round 5: one cycle ran on a box on which every process died of a GPU memory access fault and dumped core for minutes each,
# the whole budget of the round went with it): the smoke run first, nothing else on a box that fails it
if ! timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/${TAG}_canary.log 2>&1; then
  tail -5 $O/${TAG}_canary.log; echo "gpu_cycle: the smoke run failed on this box -- nothing else is started"; exit 3
fi
ulimit -c 0            # (no core dumps of processes with 100+ GB of mappings)
b() { name=$1; shift; timeout 400 python bench.py "$@" > $O/${TAG}_bench_$name.json 2> $O/${TAG}_bench_$name.err; tail -c 1800 $O/${TAG}_bench_$name.json; tail -2 $O/${TAG}_bench_$name.err; }
if has tests; then
  timeout 1500 python -m pytest tests -m gpu -x -q --durations=6 2>&1 | tail -16 | tee $O/${TAG}_pytest.log
fi
if has bench; then
  b driver_flags --steps 20 --warmup 5
  b saturated --steps 300 --warmup 150
  b quiescent --steps 300 --warmup 150 --regime quiescent --no-cpu-baseline
fi
if has extra; then
  b saturated_gc --steps 300 --warmup 150 --gc --no-cpu-baseline
  b saturated_robust --steps 300 --warmup 150 --scheme robust --no-cpu-baseline
  b saturated_p10 --steps 100 --warmup 20 --num-to-gossip 10 --no-cpu-baseline
  # the lossy lines carry the same-cluster CPU figure and the oracle check (shorter windows: the threaded oracle needs them)
  b loss1pct_gc --steps 60 --warmup 20 --loss-ppm 10000 --gc
  b loss30pct_16k --steps 200 --warmup 20 --members 16384 --loss-ppm 300000
  b loss30pct_32k --steps 100 --warmup 20 --members 32768 --loss-ppm 300000   # (65 536 members at 30 % loss: every member a subject, beyond the 65 534 view rows a handle can have)
fi
if has variants; then
  # the tick kernels with the state by value (product) against by pointer (libswimsim_sptr.so): the same cluster, stepped in turn
  timeout 600 python scripts/ab_time.py swim_amd/csrc/libswimsim.so swim_amd/csrc/libswimsim_sptr.so 2>&1 | tee $O/${TAG}_variants_state_by_pointer.txt
fi
if has shard; then
  # one population as 1 / 2 / 4 / 8 handles on this GPU, both forms of the exchange
  timeout 600 python scripts/shard_time.py 1 2 4 8 2>&1 | tee $O/${TAG}_shard_overhead_one_gpu.txt
fi
if has bounded; then
  # BASELINE config 5 at its per-GPU size with bounded member maps (view_cap; swim_sparse.h): bench lines (oracle-verified in the
  # same run), kernel times by capacity, the config's two reported numbers (false-positive Dead, ticks-to-all) x churn
  b config5_2m_cap64 --steps 20 --warmup 5 --members 2097152 --loss-ppm 300000 --view-cap 64
  b config5_2m_cap64_churn1pct --steps 20 --warmup 5 --members 2097152 --loss-ppm 300000 --view-cap 64 --churn 10
  b config5_1m_cap256 --steps 20 --warmup 5 --members 1048576 --loss-ppm 300000 --view-cap 256
  (timeout 600 python scripts/bounded_time.py 65536 64 262144 64 1048576 64 2097152 64 2097152 128 2097152 256 4194304 64; echo '# sp_probe_kernel (one wave per member), SWIMSIM_SP_PROBE=wave:'; SWIMSIM_SP_PROBE=wave timeout 300 python scripts/bounded_time.py 2097152 64) 2>&1 | grep -v amdgpu.ids | tee $O/${TAG}_bounded_time.txt
  (CAP=64 TICKS=160 T0=60 timeout 600 python scripts/config5.py 2097152; CAP=256 TICKS=200 T0=60 timeout 900 python scripts/config5.py 2097152; echo '# oracle-checked at 262 144 members:'; CAP=64 TICKS=80 T0=30 ORACLE=1 timeout 900 python scripts/config5.py 262144) 2>&1 | grep -v amdgpu.ids | tee $O/${TAG}_config5_bounded.txt
  timeout 300 python scripts/bounded_sections.py 2097152 64 2>&1 | grep -v amdgpu.ids > $O/${TAG}_bounded_sections_2m_cap64.json
fi
if has cluster; then   # BASELINE config 5 at FULL size as 8 handles on this GPU: the exchange on the handles' streams, then through the host
  (echo "# swimsim_cluster_step (exchange on the handles' streams):"; MEMBERS=4194304 SHARDS=4 timeout 600 python scripts/config5_cluster_one_gpu.py; timeout 900 python scripts/config5_cluster_one_gpu.py;
   echo "# phase calls + LocalFabric (host in the loop: SWIMSIM_CLUSTER_STEP=0):"; SWIMSIM_CLUSTER_STEP=0 timeout 900 python scripts/config5_cluster_one_gpu.py) 2>&1 | grep -v amdgpu.ids | tee $O/${TAG}_config5_cluster_one_gpu.txt
fi
if has strict; then    # strict_reference_rules: what the literal rule costs
  (echo "# strict_reference_rules, 1 M members, saturated regime:"; STRICT=1 TICKS=50 timeout 600 python scripts/quick_time.py; echo "# the default (merge) on the same cluster:"; TICKS=50 timeout 300 python scripts/quick_time.py; echo "# 1 % loss, strict / default:"; STRICT=1 LOSS=10000 TICKS=30 timeout 600 python scripts/quick_time.py; LOSS=10000 TICKS=30 timeout 600 python scripts/quick_time.py) 2>&1 | grep -v amdgpu.ids | tee $O/${TAG}_strict_time.txt
fi
cd /tmp && export TMPDIR=/tmp
if has prof; then
  # kernel-trace statistics over the TIMED WINDOW of the driver-flag line (what roofline.kernels.*.avg_launch_us must reproduce)
  bash $R/scripts/prof_timed_window.sh $TAG --steps 20 --warmup 5
fi
if has pmc; then
  PASSES="${PASSES:-p1 p2 p4}" bash $R/scripts/pmc_passes.sh $O/${TAG}_pmc > $O/${TAG}_pmc.log 2>&1; tail -40 $O/${TAG}_pmc/summary.txt
fi
if has churn; then
  ORACLE=300000 timeout 1200 python $R/scripts/churn_time.py 2>&1 | grep -v amdgpu.ids | tee $O/${TAG}_churn.txt
fi
if has config5; then   # BASELINE config 5's reported numbers (ticks-to-all, false-positive Dead): 16 384 members oracle-checked, 32 768 alone
  (ORACLE=1 timeout 900 python $R/scripts/config5.py 16384; timeout 600 python $R/scripts/config5.py 32768; echo '# 5 % loss (the protocol still converges):'; LOSS=50000 ORACLE=1 timeout 900 python $R/scripts/config5.py 16384; echo '# 1 % loss, 1 M members:'; LOSS=10000 TICKS=220 T0=100 ROWS=12000 CHURN=0,1 timeout 900 python $R/scripts/config5.py 1048576) 2>&1 | grep -v amdgpu.ids | tee $O/${TAG}_config5.txt
fi
