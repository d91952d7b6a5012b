set -x
mkdir -p gpurun_out/r4c
python -m pytest tests/test_gpu_ops.py tests/test_gpu_edges.py tests/test_gpu_pipeline.py tests/test_gpu_pipeline_kernels.py -m gpu -q -x > gpurun_out/r4c/pytest_a.log 2>&1; echo "rc=$?" >> gpurun_out/r4c/pytest_a.log
tail -6 gpurun_out/r4c/pytest_a.log
run() { tag=$1; shift; env "$@" python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r4c/bench_$tag.json 2> gpurun_out/r4c/bench_$tag.err || tail -3 gpurun_out/r4c/bench_$tag.err; }
run pp3 X=1
run pp4 PF_ATT_PP_NS=4
run nopp PF_ATT_PP=0
run pp3b X=1
python -c "
import json
for f in ('pp3','pp4','nopp','pp3b'):
    try:
        d=json.load(open('gpurun_out/r4c/bench_'+f+'.json')); c=d['class_ms_per_step']; print(f, round(d['ms_per_step'],3), d['ids_vs_fp32_oracle']['ok'], d['ids_sha1'][:8], 'ffn2',c['gemm_ffn2']['ms'],'out',c['gemm_out']['ms'],'qkv',c['gemm_qkv']['ms'],'ffn1',c['gemm_ffn1']['ms'],'attn',c['attn_self']['ms'],'ln',c['layernorm']['ms'])
    except Exception as e: print(f,'FAILED',e)
"
python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_ops.py --deselect tests/test_gpu_pipeline.py --deselect tests/test_gpu_pipeline_kernels.py > gpurun_out/r4c/pytest_b.log 2>&1; echo "rc=$?" >> gpurun_out/r4c/pytest_b.log
tail -6 gpurun_out/r4c/pytest_b.log
