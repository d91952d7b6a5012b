timeout 900 python -m pytest tests/test_gpu_baseline_sizes.py tests/test_gpu_pipeline.py -x -q 2>&1 | tail -5
for b in 0 1 0 1; do
  echo "PF_BIGP=$b"
  PF_BIGP=$b timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['ms_per_step'],3), d['ids_sha1'][:8], d['roofline']['kernel'][:40], round(d['roofline']['frac'],4), round(d['roofline']['avg_us'],2))
"
done
