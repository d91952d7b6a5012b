O=gpurun_out/r4h; mkdir -p $O
g() { tag=$1; shift; env "$@" > $O/g_$tag.json 2> $O/g_$tag.err || tail -3 $O/g_$tag.err; }
g 1x32 X=1 python bench.py --group 1 --group-devices 0 --batch 32 --steps 8 --warmup 3
g 2x16 X=1 python bench.py --group 2 --group-devices 0,0 --batch 16 --steps 8 --warmup 3
g 2x16_cap128 PF_CU_CAP=128 python bench.py --group 2 --group-devices 0,0 --batch 16 --steps 8 --warmup 3
g 2x16_cap160 PF_CU_CAP=160 python bench.py --group 2 --group-devices 0,0 --batch 16 --steps 8 --warmup 3
g 4x8_cap64 PF_CU_CAP=64 python bench.py --group 4 --group-devices 0,0,0,0 --batch 8 --steps 8 --warmup 3
g 2x32_cap128 PF_CU_CAP=128 python bench.py --group 2 --group-devices 0,0 --batch 32 --steps 6 --warmup 2
g 2x32 X=1 python bench.py --group 2 --group-devices 0,0 --batch 32 --steps 6 --warmup 2
python -c "
import json
for f in ('1x32','2x16','2x16_cap128','2x16_cap160','4x8_cap64','2x32_cap128','2x32'):
    try:
        d=json.load(open('$O/g_'+f+'.json')); print(f, 'ms/call', round(d['ms_per_step'],3), 'per 32 utt', round(d['ms_per_step']*32/d['config']['global_batch'],3), round(d['value']))
    except Exception as e: print(f,'FAILED',e)
"
