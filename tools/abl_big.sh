#!/bin/bash
# Ablation of the 256 x {192,256} GEMMs (k_gemm_big.hip): timing only, ablated variants compute garbage.  The bits are
# COMPILE-TIME (-DPF_BIG_ABL=n): run-time flags put a branch and a full lgkmcnt(0) in front of every MFMA.
# usage (on the GPU box): tools/abl_big.sh "0 1 2 4 8 3 11 15"
cd "$(dirname "$0")/.."
for a in ${1:-0 1 2 4 8}; do
  touch aliparaformerasr_amd/csrc/k_gemm_big.hip
  make -C aliparaformerasr_amd/csrc EXTRA=-DPF_BIG_ABL=$a > /dev/null 2>&1 || { echo "build failed abl=$a"; continue; }
  echo "== PF_BIG_ABL=$a"
  python tools/bench_gemm.py 2>&1 | grep -E " big|persistent" | grep -E "16000 x  (1536|2048)"
  ABL_M=64000 python tools/abl_bigp.py 2>&1 | tail -1
done
touch aliparaformerasr_amd/csrc/k_gemm_big.hip
make -C aliparaformerasr_amd/csrc > /dev/null 2>&1
