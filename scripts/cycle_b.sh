cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
WARM=150 TICKS=200 timeout 300 python scripts/quick_time.py swim_amd/csrc/libswimsim.so swim_amd/csrc/libswimsim_mw4.so swim_amd/csrc/libswimsim_mw3.so 2>&1 | tee $O/r02b_variants.txt
timeout 300 python bench.py --steps 100 --warmup 20 --gc --no-cpu-baseline 2>&1 | tail -c 700 | tee $O/r02b_gc.txt
timeout 200 python scripts/oracle_scaling.py 2>&1 | tee $O/r02b_oracle_scaling.txt
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "random_configurations or settling or forced or wraps or loss_at or default_cap" 2>&1 | tail -5 | tee $O/r02b_pytest.txt
cd /tmp; export TMPDIR=/tmp WARM=100 TICKS=40
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $GRAFT_REPO_ROOT/$O/r02b_pmc/p4 -o p -- python $GRAFT_REPO_ROOT/scripts/quick_time.py > /dev/null 2>&1
python $GRAFT_REPO_ROOT/scripts/pmc_summary.py $GRAFT_REPO_ROOT/$O/r02b_pmc 30 | tee $GRAFT_REPO_ROOT/$O/r02b_pmc_summary.txt; find $GRAFT_REPO_ROOT/$O/r02b_pmc -name "*.csv" -size +1M -delete
