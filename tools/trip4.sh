set -x
O=gpurun_out/r4d; mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -8 $O/pytest.log
python tools/bench_k32.py > $O/bench_k32.txt 2>&1; cat $O/bench_k32.txt
b() { tag=$1; shift; python bench.py --no-cpu-baseline "$@" > $O/bench_$tag.json 2> $O/bench_$tag.err || tail -3 $O/bench_$tag.err; }
b fp32 --steps 3 --warmup 1 --accuracy fp32
b int8 --steps 5 --warmup 2 --accuracy int8
b seaco_int8 --steps 5 --warmup 2 --model seaco --accuracy int8
b seaco --steps 5 --warmup 2 --model seaco
b sensevoice --steps 5 --warmup 2 --model sensevoice
b batch128 --steps 5 --warmup 2 --batch 128
python -c "
import json
for f in ('fp32','int8','seaco_int8','seaco','sensevoice','batch128'):
    try:
        d=json.load(open('$O/bench_'+f+'.json')); print(f, round(d['ms_per_step'],3), round(d['value']), d['ids_vs_fp32_oracle'], d['token_num_sum'])
    except Exception as e: print(f,'FAILED',e)
"
