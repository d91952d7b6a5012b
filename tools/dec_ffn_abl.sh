#!/bin/bash
# Compile-time ablations of the decoder (split) form of the fused FFN kernel at M = 5344, S = 3: what is the ~17 us of
# fixed time per launch?  ABL bits as in tools/ffn_abl.sh (1 = no weight loads in the main loop, 2 = no LDS fragment reads,
# 4 = no MFMA, 8 = no chunk barriers); ablated runs skip the finishing pass.   bash tools/dec_ffn_abl.sh > gpurun_out/dec_ffn_abl.txt
cd aliparaformerasr_amd/csrc && touch k_ffn.hip && make -j8 EXTRA=-DPF_FFN_ABLATIONS > /dev/null 2>&1 && cd ../..
for abl in 0 1 3 7 15; do
  echo "== ABL=$abl"
  PF_DEC_ABL=$abl PF_OP_REPEAT=17 timeout 120 python tools/bench_dec_ffn.py 5344 3 2>&1 | grep "splits"
done
cd aliparaformerasr_amd/csrc && touch k_ffn.hip && make -j8 > /dev/null 2>&1
