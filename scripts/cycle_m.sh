cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
( time timeout -s INT 420 python -X faulthandler -m pytest tests/test_hip_parity.py -m gpu -x -q --durations=5 -k "million_members_digest or golden or million_members_properties or (sharded_cluster_on_one_gpu and 4096)" ) 2>&1 | tail -60 | tee $O/r02m_slow_test.txt
