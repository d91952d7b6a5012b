#!/bin/bash
# usage (on the GPU box): tools/ffn_timing.sh -> phase timeline of the encoder's fused launch (build with -DFF_TIMING: s_memtime stamps per wave,
# k_ffn.hip; results of that build are valid, only a little slower)
cd "$(dirname "$0")/.."
touch aliparaformerasr_amd/csrc/k_ffn.hip
make -C aliparaformerasr_amd/csrc EXTRA="-DFF_TIMING $FF_EXTRA" > /dev/null 2>&1 || { echo "build failed"; exit 1; }
python ${FF_PY:-tools/ffn_timing.py}
touch aliparaformerasr_amd/csrc/k_ffn.hip
make -C aliparaformerasr_amd/csrc > /dev/null 2>&1
