timeout 600 python -m pytest tests/test_gpu_pipeline_kernels.py -x -q -k "big" 2>&1 | tail -4
timeout 300 python tools/bench_gemm.py 2>&1 | grep -i "FFN-up"
for b in 0 1 0 1; do
  echo "PF_BIGP=$b"
  PF_BIGP=$b timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['ms_per_step'],3), d['ids_sha1'][:8], round(d['roofline']['frac'],4), round(d['roofline']['avg_us'],2))
"
done
