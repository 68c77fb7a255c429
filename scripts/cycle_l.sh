cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
( time timeout 400 python -c "import torch; print(torch.zeros(1).cuda())" ) 2>&1 | tail -4 | tee $O/r02l_torch_first_import.txt
export WARM=150 TICKS=200
timeout 300 python scripts/quick_time.py swim_amd/csrc/libswimsim.so swim_amd/csrc/libswimsim_pks.so swim_amd/csrc/libswimsim_pksnf.so 2>&1 | tee $O/r02l_variants.txt
( time timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "sharded_cluster_on_one_gpu and 4096" ) 2>&1 | tail -5 | tee $O/r02l_shard_test_time.txt
