#!/bin/bash
# experimental builds of the product sources with compile-time knobs changed, for A/B timing on the GPU
# (scripts/quick_time.py lib...).  usage: mkvar.sh tag1:"-DA=1 -DB=2" tag2:"-DC" ...  -> swim_amd/csrc/libswimsim_x_<tag>.so
# (built in parallel; *.so is git-ignored but travels with gpurun)
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/swim_amd/csrc
for spec in "$@"; do
  tag=${spec%%:*}; defs=${spec#*:}; [[ "$spec" == *:* ]] || defs=""
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wl,-Bsymbolic $defs -o $C/libswimsim_x_$tag.so $C/swimsim.hip $C/swim_wire.cpp $C/swim_bridge.cpp 2>&1 | grep -E "error|warning: v" ; echo "built x_$tag ($defs)" ) &
done
wait
