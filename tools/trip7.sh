O=gpurun_out/r4g; mkdir -p $O
for w in -1 0 1 2 4 5 8; do PF_EPI_WAIT=$w python bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_w$w.json 2> $O/bench_w$w.err; done
python -c "
import json
for f in ('-1','0','1','2','4','5','8'):
    try:
        d=json.load(open('$O/bench_w'+f+'.json')); c=d['class_ms_per_step']; print('wait',f, round(d['ms_per_step'],3), d['ids_vs_fp32_oracle']['ok'], {k:c[k]['ms'] for k in ('gemm_ffn2','gemm_dec_ffn2','gemm_dec_out','gemm_vocab','gemm_cif')})
    except Exception as e: print(f,'FAILED',e)
"
