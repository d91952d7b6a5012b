#!/bin/bash
# usage (on the GPU box): tools/attn_timing.sh  -> where the waves of the self-attention kernel spend their time, per phase of the tile loop
# (build with -DATT_TIMING: s_memtime sums per wave, k_attn.hip; results of that build are valid, only a little slower)
cd "$(dirname "$0")/.."
touch aliparaformerasr_amd/csrc/k_attn.hip
make -C aliparaformerasr_amd/csrc EXTRA=-DATT_TIMING > /dev/null 2>&1 || { echo "build failed"; exit 1; }
python tools/attn_timing.py
touch aliparaformerasr_amd/csrc/k_attn.hip
make -C aliparaformerasr_amd/csrc > /dev/null 2>&1
