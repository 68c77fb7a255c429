cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
export WARM=150 TICKS=200
timeout 100 python scripts/quick_time.py 2>&1 | tee $O/r02d_variants.txt
SWIMSIM_NO_BINS=1 timeout 100 python scripts/quick_time.py 2>&1 | tee -a $O/r02d_variants.txt
SCHEME=robust timeout 100 python scripts/quick_time.py 2>&1 | tee -a $O/r02d_variants.txt
timeout 200 python scripts/oracle_scaling.py 2>&1 | tee $O/r02d_oracle_scaling.txt
