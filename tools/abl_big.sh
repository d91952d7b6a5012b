#!/bin/bash
# ablation of the 256 x {192,256} GEMM (timing only: ablated variants compute garbage)
for a in 0 1 2 4 8 3 11 15; do
  echo "== PF_BIG_ABL=$a"
  PF_BIG_ABL=$a python tools/bench_gemm.py 2>&1 | grep -E " big" | grep -E "16000 x  (1536|2048)"
done
