#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
TAG=r04fin2
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "bounded or golden" 2>&1 | tail -3 | tee $O/${TAG}_pytest_bounded.log
(timeout 600 python scripts/bounded_time.py 65536 64 262144 64 1048576 64 2097152 64 2097152 128 2097152 256 4194304 64; echo '# sp_probe_kernel (one wave per member), SWIMSIM_SP_PROBE=wave:'; SWIMSIM_SP_PROBE=wave timeout 300 python scripts/bounded_time.py 2097152 64) 2>&1 | grep -v amdgpu.ids | tee $O/${TAG}_bounded_time.txt
b() { name=$1; shift; timeout 400 python bench.py "$@" > $O/${TAG}_bench_$name.json 2> $O/${TAG}_bench_$name.err; tail -c 600 $O/${TAG}_bench_$name.json; }
b config5_2m_cap64 --steps 20 --warmup 5 --members 2097152 --loss-ppm 300000 --view-cap 64
b config5_1m_cap256 --steps 20 --warmup 5 --members 1048576 --loss-ppm 300000 --view-cap 256
timeout 300 python scripts/bounded_sections.py 2097152 64 2>&1 | grep -v amdgpu.ids > $O/${TAG}_bounded_sections_2m_cap64.json
cd /tmp && export TMPDIR=/tmp
PASSES="p1 p2 p4" bash $R/scripts/pmc_passes.sh $O/${TAG}_pmc > $O/${TAG}_pmc.log 2>&1; tail -12 $O/${TAG}_pmc/summary.txt
