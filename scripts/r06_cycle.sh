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
esac
