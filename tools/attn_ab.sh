#!/bin/bash
# usage (on the GPU box): tools/attn_ab.sh "FLAGS_A" "FLAGS_B" ...  -> attn_self / attn_cross class times of bench.py for builds of k_attn.hip
# with the given extra compiler flags (e.g. "-DATT_DMA=0" "-DATT_DMA=1"); each variant twice, interleaved
cd "$(dirname "$0")/.."
for rep in 1 2; do
for f in "$@"; do
  touch aliparaformerasr_amd/csrc/k_attn.hip
  make -C aliparaformerasr_amd/csrc EXTRA="$f" > /dev/null 2>&1 || { echo "build failed: $f"; continue; }
  python bench.py --no-cpu-baseline --no-via-recognizer --steps 10 --warmup 3 --in-flight 1 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); b=d['class_ms_per_step']; print('%-28s attn_self %.3f ms  attn_cross %.3f ms  step %.3f ms  ids ok %s' % ('$f', b['attn_self']['ms'], b['attn_cross']['ms'], d['ms_per_step'], d['ids_vs_fp32_oracle']['ok']))"
done
done
touch aliparaformerasr_amd/csrc/k_attn.hip
make -C aliparaformerasr_amd/csrc > /dev/null 2>&1
