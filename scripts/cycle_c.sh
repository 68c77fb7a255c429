cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
export WARM=150 TICKS=200
timeout 300 python scripts/quick_time.py swim_amd/csrc/libswimsim_mw4.so swim_amd/csrc/libswimsim_m2.so swim_amd/csrc/libswimsim_m4.so swim_amd/csrc/libswimsim_m4w4.so 2>&1 | tee $O/r02c_variants.txt
SWIMSIM_NO_BINS=1 timeout 100 python scripts/quick_time.py swim_amd/csrc/libswimsim_mw4.so 2>&1 | tee -a $O/r02c_variants.txt
rocprofv3 -L 2>/dev/null | grep -oE "\b(TA|TCP|TD|TCC|SQ|SQC|GRBM)_[A-Za-z0-9_]+" | sort -u > $O/r02c_counters.txt; wc -l $O/r02c_counters.txt
cd /tmp; export TMPDIR=/tmp WARM=100 TICKS=30
for c in TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_WAVEFRONTS_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum; do
  timeout 120 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/$O/r02c_pmc/$c -o p -- python $GRAFT_REPO_ROOT/scripts/quick_time.py $GRAFT_REPO_ROOT/swim_amd/csrc/libswimsim_m2.so > /dev/null 2>&1 || echo "counter $c failed"
done
python $GRAFT_REPO_ROOT/scripts/pmc_summary.py $GRAFT_REPO_ROOT/$O/r02c_pmc 25 | grep -v "begin_kernel\|digest" | tee $GRAFT_REPO_ROOT/$O/r02c_pmc_summary.txt; find $GRAFT_REPO_ROOT/$O/r02c_pmc -name "*.csv" -size +500k -delete
