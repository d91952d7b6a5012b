#!/bin/bash
# usage (GPU box): tools/gemm_st.sh "0 1 2 0 1 2" -> GEMM class times per cache policy of the f16 result stores
cd "$(dirname "$0")/.."
for a in $1; do
  touch aliparaformerasr_amd/csrc/k_gemm.hip
  make -C aliparaformerasr_amd/csrc EXTRA=-DPF_GEMM_ST=$a > /dev/null 2>&1 || { echo "build failed st=$a"; continue; }
  python bench.py --breakdown --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); b=d['kernel_breakdown_ms_per_step']; print('ST=$a total %.2f' % d['ms_per_step'], {k[5:]: round(v['ms'],3) for k,v in b.items() if k in ('gemm_qkv','gemm_ffn1','gemm_ffn2','gemm_out','attn_self','fsmn','layernorm')})"
done
touch aliparaformerasr_amd/csrc/k_gemm.hip
make -C aliparaformerasr_amd/csrc > /dev/null 2>&1
