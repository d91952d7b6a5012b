#!/bin/bash
# Compile-time ablations of the fused FFN kernel (k_ffn.hip): which resource bounds its main loop?  Builds the library with
# -DPF_FFN_ABLATIONS, times the variants through tools/bench_ffn.py (garbage results, only the times matter), rebuilds clean.
#   bash tools/ffn_abl.sh > gpurun_out/ffn_abl.txt
cd aliparaformerasr_amd/csrc && touch k_ffn.hip && make -j8 EXTRA=-DPF_FFN_ABLATIONS > /dev/null 2>&1 && cd ../..
for abl in 0 1 2 4 3 8 7 15; do
  echo "== ABL=$abl (1 = no weight loads, 2 = no LDS fragment reads, 4 = no MFMA, 8 = no chunk barriers)"
  PF_FFN_ABL=$abl timeout 120 python tools/bench_ffn.py 2>&1 | grep "fused"
done
cd aliparaformerasr_amd/csrc && touch k_ffn.hip && make -j8 > /dev/null 2>&1
