set -x
O=gpurun_out/r4f; mkdir -p $O
python -m pytest tests/test_gpu_pipeline_kernels.py tests/test_gpu_int8.py tests/test_gpu_pipeline.py tests/test_gpu_seaco.py tests/test_gpu_interference.py -m gpu -q -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
for i in 1 2; do python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_$i.json 2> $O/bench_$i.err; done
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --accuracy int8 > $O/bench_int8.json 2> $O/bench_int8.err
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --model sensevoice > $O/bench_sv.json 2> $O/bench_sv.err
python -c "
import json
for f in ('1','2','int8','sv'):
    try:
        d=json.load(open('$O/bench_'+f+'.json')); c=d['class_ms_per_step']; print(f, round(d['ms_per_step'],3), d['ids_vs_fp32_oracle'] and d['ids_vs_fp32_oracle']['ok'], {k:v['ms'] for k,v in c.items() if k.startswith('gemm') or k in ('attn_self','layernorm')})
    except Exception as e: print(f,'FAILED',e)
"
