#!/bin/bash
# usage (GPU box): tools/fsmn_rows.sh "8 4 16"  -> fsmn class time per variant of FS_ROWS
cd "$(dirname "$0")/.."
for a in $1; do
  touch aliparaformerasr_amd/csrc/k_misc.hip
  make -C aliparaformerasr_amd/csrc EXTRA="-DFS_ST=$a" > /dev/null 2>&1 || { echo "build failed rows=$a"; continue; }
  python bench.py --breakdown --no-cpu-baseline --steps 2 --warmup 1 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); b=d['kernel_breakdown_ms_per_step']; print('FS_ST=$a fsmn %.3f attn %.3f out %.3f total %.2f ms' % (b['fsmn']['ms'], b['attn_self']['ms'], b['gemm_out']['ms'], d['ms_per_step']))"
done
touch aliparaformerasr_amd/csrc/k_misc.hip
make -C aliparaformerasr_amd/csrc > /dev/null 2>&1
