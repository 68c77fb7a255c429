#!/bin/bash
# Collect PMC counters for the loaded 1M-member regime in separate passes (TCC has 4 slots:
# FETCH_SIZE costs 3, WRITE_SIZE 2).  Usage: pmc_passes.sh <outdir> <per_mille>
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$1; PM=${2:-1}
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp WARM=${WARM:-150} TICKS=${TICKS:-60}
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/p1 -o p -- python $R/scripts/quick_time.py $PM > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/p2 -o p -- python $R/scripts/quick_time.py $PM > $OUT/p2.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_REQ_sum TCC_ATOMIC_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d $OUT/p3 -o p -- python $R/scripts/quick_time.py $PM > $OUT/p3.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/p4 -o p -- python $R/scripts/quick_time.py $PM > $OUT/p4.log 2>&1
