#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
TAG=r04n
(timeout 600 python scripts/ab_time.py swim_amd/csrc/libswimsim_x_old.so swim_amd/csrc/libswimsim.so; echo "# 1 % loss:"; LOSS=10000 ROUNDS=5 CHUNK=20 timeout 600 python scripts/ab_time.py swim_amd/csrc/libswimsim_x_old.so swim_amd/csrc/libswimsim.so) 2>&1 | grep -v amdgpu.ids | tee $O/${TAG}_ab_deadline_batches.txt
TICKS=100 timeout 300 python scripts/section_clocks.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_section_clocks_saturated.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04n_section_clocks_saturated.json'))
print(d['merge_us'], d['merge_kernel']['clocks_per_wave'])
for k,v in d['merge_kernel']['sections'].items(): print('  %-50s %8.0f %.3f'%(k,v['clocks_per_wave'],v['share']))
PY
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "config1 or config2 or small_pop or saturated or random or churn or loss" 2>&1 | tail -4
