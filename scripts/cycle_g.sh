cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
export WARM=150 TICKS=200
timeout 300 python scripts/quick_time.py swim_amd/csrc/libswimsim.so swim_amd/csrc/libswimsim_vt0.so swim_amd/csrc/libswimsim_vt1024.so 2>&1 | tee $O/r02g_variants.txt
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $O/r02g_pytest.log
for a in "loss1pct_gc --steps 100 --warmup 30 --loss-ppm 10000 --gc" "loss30pct_16k --steps 100 --warmup 20 --members 16384 --loss-ppm 300000" "loss5pct_64k_gc --steps 100 --warmup 20 --members 65536 --loss-ppm 50000 --gc"; do set -- $a; n=$1; shift; timeout 300 python bench.py "$@" --no-cpu-baseline > $O/r02g_bench_$n.json 2> $O/r02g_bench_$n.err; python -c "
import json,sys; j=json.loads(open('$O/r02g_bench_$n.json').read().strip().splitlines()[-1]); print('$n', '%.3g'%j['value'], round(j['ms_per_step']*1000,1), {k:round(v['avg_launch_us'],1) for k,v in j['roofline']['kernels'].items()}, j['per_member_tick'])" 2>&1 | tail -1; tail -1 $O/r02g_bench_$n.err; done
cd /tmp; export TMPDIR=/tmp WARM=100 TICKS=30
for c in TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_PERMISSION_MISS_sum; do
  timeout 120 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/$O/r02g_pmc/$c -o p -- python $GRAFT_REPO_ROOT/scripts/quick_time.py > /dev/null 2>&1 || echo "counter $c failed"
done
python $GRAFT_REPO_ROOT/scripts/pmc_summary.py $GRAFT_REPO_ROOT/$O/r02g_pmc 25 | grep -v "begin_kernel\|digest" | tee $GRAFT_REPO_ROOT/$O/r02g_pmc_summary.txt; find $GRAFT_REPO_ROOT/$O/r02g_pmc -name "*.csv" -size +500k -delete
grep -i "utcl\|tlb" $GRAFT_REPO_ROOT/$O/r02c_counters.txt 2>/dev/null | head -20
