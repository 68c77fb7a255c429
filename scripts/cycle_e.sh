cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
export WARM=150 TICKS=200
timeout 300 python scripts/quick_time.py swim_amd/csrc/libswimsim.so swim_amd/csrc/libswimsim_gb1.so swim_amd/csrc/libswimsim_gb2.so swim_amd/csrc/libswimsim_gb8.so 2>&1 | tee $O/r02e_variants.txt
