R=$PWD; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -E "ICACHE|IFETCH|INST_CACHE|SQC_" | head -40 > $O/r03m_counters_list.txt
run() { n=$1; lib=$2; shift; shift; WARM=230 TICKS=60 timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmc_$n -o p -- python $R/scripts/quick_time.py $R/swim_amd/csrc/$lib > /tmp/pmc_$n.log 2>&1; tail -2 /tmp/pmc_$n.log | cut -c1-300; }
for lib in libswimsim_x_base.so libswimsim_x_rp.so libswimsim_x_rk_nolaunch.so; do
  run a_$lib $lib SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_IFETCH SQ_INSTS_VALU
  run b_$lib $lib SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU
done
python $R/scripts/pmc_summary.py /tmp 40 > $O/r03m_pmc_icache.txt 2>&1
cat $O/r03m_counters_list.txt | head -30; cat $O/r03m_pmc_icache.txt
