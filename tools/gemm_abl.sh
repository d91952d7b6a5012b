#!/bin/bash
# usage (on the GPU box): tools/gemm_abl.sh N K f16out "0 1 2 4 ..."
cd "$(dirname "$0")/.."
for a in $4; do
  touch aliparaformerasr_amd/csrc/k_gemm.hip
  make -C aliparaformerasr_amd/csrc EXTRA=-DPF_ABL=$a > /dev/null 2>&1 || { echo "build failed abl=$a"; continue; }
  python tools/gemm_abl.py $a $1 $2 $3 2>&1 | grep abl=
done
touch aliparaformerasr_amd/csrc/k_gemm.hip
make -C aliparaformerasr_amd/csrc > /dev/null 2>&1
