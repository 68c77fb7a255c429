cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "wraps or config5 or default_cap or random_conf or forced or settling" 2>&1 | tail -5 | tee $O/r02j_pytest.log
WARM=150 TICKS=200 timeout 100 python scripts/quick_time.py 2>&1 | tee $O/r02j_variants.txt
