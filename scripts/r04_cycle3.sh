#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
TAG=r04c
timeout 300 python scripts/bounded_sections.py 2097152 64 2>&1 | grep -v amdgpu.ids | tee $O/${TAG}_bounded_sections_2m_cap64.json
timeout 300 python scripts/bounded_sections.py 2097152 256 2>&1 | grep -v amdgpu.ids | tee $O/${TAG}_bounded_sections_2m_cap256.json
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d /tmp/pmcA -o p -- python $R/scripts/bounded_time.py 2097152 64 > /tmp/pmcA.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU --output-format csv -d /tmp/pmcB -o p -- python $R/scripts/bounded_time.py 2097152 64 > /tmp/pmcB.log 2>&1
python $R/scripts/pmc_summary.py /tmp/pmcA 10 2>&1 | tee $O/${TAG}_bounded_pmc.txt; python $R/scripts/pmc_summary.py /tmp/pmcB 10 2>&1 | tee -a $O/${TAG}_bounded_pmc.txt
tail -3 /tmp/pmcA.log
