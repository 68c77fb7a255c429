cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
( time timeout 900 python -m pytest tests -m gpu -x -q --durations=5 ) 2>&1 | tail -14 | tee $O/r02n_pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/r02n_smoke.txt
