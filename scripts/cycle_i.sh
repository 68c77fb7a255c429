cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
export WARM=150 TICKS=200
timeout 200 python scripts/quick_time.py swim_amd/csrc/libswimsim.so swim_amd/csrc/libswimsim_nomb.so 2>&1 | tee $O/r02i_variants.txt
SCHEME=robust timeout 100 python scripts/quick_time.py swim_amd/csrc/libswimsim.so swim_amd/csrc/libswimsim_nomb.so 2>&1 | tee -a $O/r02i_variants.txt
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "wraps or settling or forced or config5 or default_cap or random_conf" 2>&1 | tail -5 | tee $O/r02i_pytest.log
