#!/bin/bash
# round-6 GPU cycles (one gpurun call each): r06_cycle.sh <name>; output under gpurun_out/r06<name>_*
# Every cycle starts with the smoke run (gpu_cycle.sh's canary, or the one below): nothing else is started on a box that fails it.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
C=swim_amd/csrc
canary() { if ! timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/r06$1_canary.log 2>&1; then tail -5 $O/r06$1_canary.log; echo "r06_cycle: the smoke run failed on this box -- nothing else is started"; exit 3; fi; ulimit -c 0; }
case "$1" in
a)  # the two patches of round 5 (a message from outside opens its row on every shard; the known-ring learns what a member states): suite, bench lines, trace, PMC
  bash scripts/gpu_cycle.sh r06a tests bench prof pmc
  ;;
b)  # merge_kernel: decide (wave-uniform union walk, coalesced rows) + book (per-lane, from LDS) against the per-lane walk of rounds 2-5,
    # interleaved on one cluster, in both regimes (the headline's 1 crash per tick; BASELINE.md row 3(s) as written: 9.5 per tick);
    # the guide's grid barriers
  canary b
  (echo "# headline regime (1 crash per tick): per-lane walk | union walk, batches of 4 | of 2 | of 8"; ROUNDS=7 timeout 900 python scripts/ab_time.py $C/libswimsim_x_nounion.so $C/libswimsim.so $C/libswimsim_x_u2.so $C/libswimsim_x_u8.so;
   echo "# BASELINE.md row 3(s) as written (9.5 crashes per tick, settling, 8192 rows):"; CPT=9.5 GC=1 MAXSUBJ=8192 WARM=200 CHUNK=20 ROUNDS=5 timeout 900 python scripts/ab_time.py $C/libswimsim_x_nounion.so $C/libswimsim.so $C/libswimsim_x_u2.so $C/libswimsim_x_u8.so;
   echo "# 1 % loss:"; LOSS=10000 GC=1 ROUNDS=5 CHUNK=20 timeout 900 python scripts/ab_time.py $C/libswimsim_x_nounion.so $C/libswimsim.so) 2>&1 | grep -v amdgpu.ids | tee $O/r06b_ab_merge_union.txt
  (cd scripts/microbench && timeout 300 ./grid_barrier_xcd 20 50) 2>&1 | tee $O/r06b_microbench_grid_barrier.txt
  ;;
c)  # the ring directory (a stated rumour takes id and subject from the tick's ring: no find_rid / subject_of / minfo chain) A/B; the union walk in
    # BASELINE.md row 3(s) as written; the default bench run with its second window
  canary c
  (echo "# headline regime: without | with the ring directory | union walk + directory"; ROUNDS=7 timeout 900 python scripts/ab_time.py $C/libswimsim_x_nodir.so $C/libswimsim.so $C/libswimsim_x_union.so;
   echo "# 1 % loss, settling:"; LOSS=10000 GC=1 ROUNDS=5 CHUNK=20 timeout 900 python scripts/ab_time.py $C/libswimsim_x_nodir.so $C/libswimsim.so;
   echo "# BASELINE.md row 3(s) as written (9.5 crashes per tick, settling, 8192 rows): per-lane walk | union walk"; CPT=9.5 GC=1 MAXSUBJ=8192 WARM=200 CHUNK=20 ROUNDS=5 timeout 900 python scripts/ab_time.py $C/libswimsim.so $C/libswimsim_x_union.so;
   echo "# ... without | with the ring directory"; CPT=9.5 GC=1 MAXSUBJ=8192 WARM=200 CHUNK=20 ROUNDS=5 timeout 900 python scripts/ab_time.py $C/libswimsim_x_nodir.so $C/libswimsim.so) 2>&1 | grep -v amdgpu.ids | tee $O/r06c_ab_ring_directory.txt
  timeout 900 python bench.py --steps 20 --warmup 5 > $O/r06c_bench_default.json 2> $O/r06c_bench_default.err; tail -c 3000 $O/r06c_bench_default.json; tail -3 $O/r06c_bench_default.err
  ;;
d)  # merge_kernel with ONE round of input loads (the known-ring among them) and the first rumour batch's cells asked for with the deadline
    # cells, the own line kept in registers -- against the kernels of the commit before (x_base); the whole GPU suite
  canary d
  (echo "# headline regime: before | after"; ROUNDS=7 timeout 900 python scripts/ab_time.py $C/libswimsim_x_base.so $C/libswimsim.so;
   echo "# 1 % loss, settling:"; LOSS=10000 GC=1 ROUNDS=5 CHUNK=20 timeout 900 python scripts/ab_time.py $C/libswimsim_x_base.so $C/libswimsim.so;
   echo "# BASELINE.md row 3(s) as written:"; CPT=9.5 GC=1 MAXSUBJ=8192 WARM=200 CHUNK=20 ROUNDS=5 timeout 900 python scripts/ab_time.py $C/libswimsim_x_base.so $C/libswimsim.so) 2>&1 | grep -v amdgpu.ids | tee $O/r06d_ab_merge_rounds.txt
  timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $O/r06d_pytest.log
  ;;
e)  # probe_kernel with the first draws' bytes and the own queue mask in the round of loads of the member's own word (x_base: the kernels of
    # two commits before: neither this nor merge_kernel's single round of inputs); the GPU suite
  canary e
  (echo "# headline regime: before | after"; ROUNDS=7 timeout 900 python scripts/ab_time.py $C/libswimsim_x_base.so $C/libswimsim.so;
   echo "# quiescent-like (numToGossip 10):"; P=10 ROUNDS=5 CHUNK=20 timeout 900 python scripts/ab_time.py $C/libswimsim_x_base.so $C/libswimsim.so;
   echo "# 1 % loss, settling:"; LOSS=10000 GC=1 ROUNDS=5 CHUNK=20 timeout 900 python scripts/ab_time.py $C/libswimsim_x_base.so $C/libswimsim.so) 2>&1 | grep -v amdgpu.ids | tee $O/r06e_ab_probe_first_draws.txt
  timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $O/r06e_pytest.log
  ;;
f)  # where the waves of the two tick kernels spend their time now (section clocks); binned pushes against global atomics; the new tests
  canary f
  timeout 600 python scripts/section_clocks.py 2>/dev/null | grep -v amdgpu.ids > $O/r06f_section_clocks_saturated.json; python -c "
import json; d=json.load(open('$O/r06f_section_clocks_saturated.json'))
for k in ('merge_kernel','probe_kernel'):
    print(k, d[k]['clocks_per_wave'], {n: v['clocks_per_wave'] for n, v in d[k]['sections'].items() if v['clocks_per_wave'] > 100})
print(d['events_per_tick'], d['probe_us'], d['merge_us'])"
  (cd scripts/microbench && timeout 300 ./push_binning 20 50) 2>&1 | tee $O/r06f_microbench_push_binning.txt
  timeout 1200 python -m pytest tests -m gpu -x -q -k "event_stream or bridge or golden" 2>&1 | tail -5 | tee $O/r06f_pytest_subset.log
  ;;
g)  # the slow regimes: explicit records as a phase of merge_kernel against records_kernel in BASELINE.md row 3(s) as written; where the waves go there and at 1 % loss
  canary g
  (echo "# BASELINE.md row 3(s) as written: records as a phase of merge_kernel (the lossless handle's choice) | records_kernel"; CPT=9.5 GC=1 MAXSUBJ=8192 WARM=200 CHUNK=20 ROUNDS=5 timeout 900 python scripts/ab_time.py $C/libswimsim.so@SWIMSIM_RECORDS_KERNEL=0 $C/libswimsim.so@SWIMSIM_RECORDS_KERNEL=1) 2>&1 | grep -v amdgpu.ids | tee $O/r06g_ab_records_kernel_as_written.txt
  CPT=9.5 GC=1 MAXSUBJ=8192 WARM=200 TICKS=40 timeout 600 python scripts/section_clocks.py 2>/dev/null | grep -v amdgpu.ids > $O/r06g_section_clocks_as_written.json
  LOSS=10000 GC=1 WARM=150 TICKS=40 timeout 600 python scripts/section_clocks.py 2>/dev/null | grep -v amdgpu.ids > $O/r06g_section_clocks_loss1pct.json
  for f in as_written loss1pct; do python -c "
import json; d=json.load(open('$O/r06g_section_clocks_$f.json'))
for k in ('merge_kernel','probe_kernel','records_kernel'):
    print('$f', k, d[k]['clocks_per_wave'], {n: v['clocks_per_wave'] for n, v in d[k]['sections'].items() if v['clocks_per_wave'] > 300})
print(d['events_per_tick'], d['probe_us'], d['merge_us'])"; done
  ;;
fin)  # the cycle of the round's kernels: the GPU suite, the bench lines (the driver's flags first: both windows), kernel trace over the timed
      # window, PMC traffic, the loss / robust / P=10 / settling lines, shards on one GPU, config 4 at full size, config 5 per GPU
  bash scripts/gpu_cycle.sh r06fin tests bench extra prof pmc
  for G in 2 4; do SWIM_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus $G --steps 20 --warmup 5 2>/dev/null | grep -v amdgpu.ids > $O/r06fin_bench_gpus${G}_one_process_shared_gpu.json; done
  (echo "# strong: 1 048 576 members as G handles"; FORMS=cluster,phases KERNELS=1 timeout 900 python scripts/shard_time.py 1 2 4 8;
   for G in 2 4; do echo "# weak: $G x 1 048 576 members vs one handle of $((G * 1048576))"; MEMBERS=$((G * 1048576)) FORMS=cluster WARM=100 TICKS=30 timeout 900 python scripts/shard_time.py 1 $G; done) 2>&1 | grep -v amdgpu.ids | tee $O/r06fin_shard_overhead_one_gpu.txt
  timeout 1200 python scripts/config4_one_gpu.py 2>&1 | grep -v amdgpu.ids | tee $O/r06fin_config4_one_gpu.txt
  timeout 600 python bench.py --steps 20 --warmup 5 --members 2097152 --loss-ppm 300000 --view-cap 64 > $O/r06fin_bench_config5_2m_cap64.json 2> $O/r06fin_bench_config5_2m_cap64.err; tail -c 600 $O/r06fin_bench_config5_2m_cap64.json
  ;;
fin2)  # the kernel trace over the timed window of BOTH windows of the driver's line (the first fin cycle traced the second window only), the
       # driver's line once more with traffic.json in place, a random parity sweep on the round's kernels
  canary fin2
  (cd /tmp && export TMPDIR=/tmp && bash $R/scripts/prof_timed_window.sh r06fin --steps 20 --warmup 5 && AS_WRITTEN=1 bash $R/scripts/prof_timed_window.sh r06fin_as_written --steps 20 --warmup 5)
  timeout 900 python bench.py --steps 20 --warmup 5 > $O/r06fin_bench_driver_flags.json 2> $O/r06fin_bench_driver_flags.err; tail -c 400 $O/r06fin_bench_driver_flags.json
  timeout 1500 python scripts/gpu_parity_sweep.py 2>&1 | grep -v amdgpu.ids | tail -15 | tee $O/r06fin_gpu_parity_sweep.txt
  ;;
fin3)  # the round's LAST sources (records-mode hysteresis; comments renumbered: the sha of the kernel sources changed): suite, the driver's
       # line, both traces, PMC passes -> traffic.json is re-keyed from this run
  bash scripts/gpu_cycle.sh r06fin3 tests pmc
  (cd /tmp && export TMPDIR=/tmp && bash $R/scripts/prof_timed_window.sh r06fin3 --steps 20 --warmup 5 && AS_WRITTEN=1 bash $R/scripts/prof_timed_window.sh r06fin3_as_written --steps 20 --warmup 5)
  python scripts/make_traffic_json.py $O/r06fin3_pmc/summary.txt r06fin3 > /dev/null && cp profiles/traffic.json $O/r06fin3_traffic.json
  timeout 900 python bench.py --steps 20 --warmup 5 > $O/r06fin3_bench_driver_flags.json 2> $O/r06fin3_bench_driver_flags.err; tail -c 300 $O/r06fin3_bench_driver_flags.json
  timeout 900 python bench.py > $O/r06fin3_bench_default_flags.json 2> $O/r06fin3_bench_default_flags.err; tail -c 300 $O/r06fin3_bench_default_flags.json
  ;;
h)  # batch sizes of merge_kernel's three loops of dependent view-cell loads (deadline cells 4 / 7, todo entries 2 / 4, rumour positions 2 / 4) and
    # 3 waves per SIMD with all of them large: the headline, 1 % loss, row 3(s) as written (two handles at a time: 64 GB of view rows each)
  canary h
  L="$C/libswimsim_x_db4.so $C/libswimsim_x_db7.so $C/libswimsim_x_tb4.so $C/libswimsim_x_gb4.so $C/libswimsim_x_db7tb4.so $C/libswimsim_x_w3all.so"
  (echo "# headline regime: product (deadline batch 4, todo 2, rumours 2) | deadline 7 | todo 4 | rumours 4 | deadline 7 + todo 4 | 3 waves, 7 / 4 / 4"; ROUNDS=7 timeout 900 python scripts/ab_time.py $L;
   echo "# 1 % loss, settling:"; LOSS=10000 GC=1 ROUNDS=5 CHUNK=20 timeout 900 python scripts/ab_time.py $L;
   for v in db7 tb4 db7tb4 w3all; do echo "# BASELINE.md row 3(s) as written: product | $v"; CPT=9.5 GC=1 MAXSUBJ=8192 WARM=200 CHUNK=20 ROUNDS=5 timeout 900 python scripts/ab_time.py $C/libswimsim_x_db4.so $C/libswimsim_x_$v.so; done) 2>&1 | grep -v amdgpu.ids | tee $O/r06h_ab_merge_batches.txt
  ;;
fin4)  # (superseded by fin5) after a source change (the literal rule with settling and state pulls: swimsim.hip's configuration check): suite, PMC,
       # traces of both windows, traffic.json re-keyed, the driver's line, a random parity sweep with the new combinations
  bash scripts/gpu_cycle.sh r06fin4 tests pmc
  (cd /tmp && export TMPDIR=/tmp && bash $R/scripts/prof_timed_window.sh r06fin4 --steps 20 --warmup 5 && AS_WRITTEN=1 bash $R/scripts/prof_timed_window.sh r06fin4_as_written --steps 20 --warmup 5)
  python scripts/make_traffic_json.py $O/r06fin4_pmc/summary.txt r06fin4 > /dev/null && cp profiles/traffic.json $O/r06fin4_traffic.json
  timeout 900 python bench.py --steps 20 --warmup 5 > $O/r06fin4_bench_driver_flags.json 2> $O/r06fin4_bench_driver_flags.err; tail -c 300 $O/r06fin4_bench_driver_flags.json
  SEED=777 timeout 1500 python scripts/gpu_parity_sweep.py 2>&1 | grep -v amdgpu.ids | tail -8 | tee $O/r06fin4_gpu_parity_sweep.txt
  ;;
fin5)  # the round's LAST kernel sources (young_rid's bound, the own-known test by allocation number): suite, PMC, traces of both windows,
       # traffic.json re-keyed, the driver's line and the default line, a random parity sweep
  bash scripts/gpu_cycle.sh r06fin5 tests pmc
  (cd /tmp && export TMPDIR=/tmp && bash $R/scripts/prof_timed_window.sh r06fin5 --steps 20 --warmup 5 && AS_WRITTEN=1 bash $R/scripts/prof_timed_window.sh r06fin5_as_written --steps 20 --warmup 5)
  python scripts/make_traffic_json.py $O/r06fin5_pmc/summary.txt r06fin5 > /dev/null && cp profiles/traffic.json $O/r06fin5_traffic.json
  timeout 900 python bench.py --steps 20 --warmup 5 > $O/r06fin5_bench_driver_flags.json 2> $O/r06fin5_bench_driver_flags.err; tail -c 300 $O/r06fin5_bench_driver_flags.json
  timeout 900 python bench.py > $O/r06fin5_bench_default_flags.json 2> $O/r06fin5_bench_default_flags.err; tail -c 300 $O/r06fin5_bench_default_flags.json
  SEED=4242 timeout 1500 python scripts/gpu_parity_sweep.py 2>&1 | grep -v amdgpu.ids | tail -4 | tee $O/r06fin5_gpu_parity_sweep.txt
  ;;
esac
