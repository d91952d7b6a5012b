timeout 600 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_baseline_sizes.py tests/test_gpu_seaco.py -x -q 2>&1 | tail -5
for f in 0 1 2 4 7; do
  echo "PF_DEC_FUSE=$f"
  PF_DEC_FUSE=$f python bench.py --steps 10 --warmup 3 --no-cpu-baseline --breakdown 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
kb=d['kernel_breakdown_ms_per_step']
print(round(d['ms_per_step'],3), d['ids_sha1'][:8], {k:round(v['ms'],3) for k,v in kb.items() if ('dec' in k or k in ('layernorm','fsmn','attn_cross')) })
"
done
