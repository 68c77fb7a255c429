#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
TAG=r04fin7
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee $O/${TAG}_pytest.log
timeout 200 python bench.py --steps 300 --warmup 150 --regime quiescent --no-cpu-baseline > $O/${TAG}_bench_quiescent.json 2> $O/${TAG}_bench_quiescent.err; tail -c 300 $O/${TAG}_bench_quiescent.json
SWIMSIM_FOLD_BEGIN=0 timeout 200 python bench.py --steps 300 --warmup 150 --regime quiescent --no-cpu-baseline > $O/${TAG}_bench_quiescent_nofold.json 2>/dev/null
python - <<'PY'
import json
for n in ("quiescent","quiescent_nofold"):
    d=json.loads(open("gpurun_out/r04fin7_bench_%s.json"%n).read().strip().splitlines()[-1]); print(n, d["ms_per_step"], d["value"])
PY
cd /tmp && export TMPDIR=/tmp
PASSES="p1 p2" bash $R/scripts/pmc_passes.sh $O/${TAG}_pmc > $O/${TAG}_pmc.log 2>&1; tail -8 $O/${TAG}_pmc/summary.txt | cut -c1-160
