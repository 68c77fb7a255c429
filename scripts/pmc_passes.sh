#!/bin/bash
# PMC counters for the saturated 1M-member regime, one rocprofv3 run per counter group (TCC has few
# slots: FETCH_SIZE and WRITE_SIZE go in their own passes; never combined with other trace domains).
# Usage: pmc_passes.sh <outdir>
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$1
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp WARM=${WARM:-150} TICKS=${TICKS:-40}
# PASSES="p1 p2 p4" limits the passes (default: all)
run() { n=$1; shift; [[ -n "$PASSES" && " $PASSES " != *" $n "* ]] && return; timeout 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$n -o p -- python $R/scripts/quick_time.py > $OUT/$n.log 2>&1; }
run p1 FETCH_SIZE
run p2 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
run p3 TCC_REQ_sum TCC_ATOMIC_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
run p4 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run p5 SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
python $R/scripts/pmc_summary.py $OUT 30 > $OUT/summary.txt 2>&1; cat $OUT/summary.txt
# keep only the summary and logs (the raw per-dispatch CSVs are tens of MB)
find $OUT -name "*.csv" -size +2M -delete
