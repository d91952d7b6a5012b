set -x
mkdir -p gpurun_out/r4b
python -m pytest tests -m gpu -q -x > gpurun_out/r4b/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4b/pytest.log
tail -8 gpurun_out/r4b/pytest.log
run() { tag=$1; shift; env "$@" python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r4b/bench_$tag.json 2> gpurun_out/r4b/bench_$tag.err || tail -3 gpurun_out/r4b/bench_$tag.err; }
run new X=1
run nok32 PF_K32=0
run norcpre PF_RC_PRE=0
run old PF_K32=0 PF_RC_PRE=0
run new2 X=1
python -c "
import json
for f in ('new','nok32','norcpre','old','new2'):
    try:
        d=json.load(open('gpurun_out/r4b/bench_'+f+'.json')); c=d['class_ms_per_step']; print(f, round(d['ms_per_step'],3), d['ids_vs_fp32_oracle']['ok'], d['ids_sha1'][:8], 'ffn2',c['gemm_ffn2']['ms'],c['gemm_ffn2']['kernel'],'out',c['gemm_out']['ms'],'qkv',c['gemm_qkv']['ms'],'ffn1',c['gemm_ffn1']['ms'],'attn',c['attn_self']['ms'],'ln',c['layernorm']['ms'])
    except Exception as e: print(f,'FAILED',e)
"
