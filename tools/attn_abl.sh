#!/bin/bash
# usage (on the GPU box): tools/attn_abl.sh "0 1 2 ..."   -> per-variant attn_self time from bench --breakdown
cd "$(dirname "$0")/.."
for a in $1; do
  touch aliparaformerasr_amd/csrc/k_attn.hip
  make -C aliparaformerasr_amd/csrc EXTRA=-DATT_ABL=$a > /dev/null 2>&1 || { echo "build failed abl=$a"; continue; }
  python bench.py --breakdown --no-cpu-baseline --steps 1 --warmup 1 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); b=d['kernel_breakdown_ms_per_step']; print('abl=$a attn_self %.3f ms  attn_cross %.3f ms' % (b['attn_self']['ms'], b['attn_cross']['ms']))"
done
touch aliparaformerasr_amd/csrc/k_attn.hip
make -C aliparaformerasr_amd/csrc > /dev/null 2>&1
