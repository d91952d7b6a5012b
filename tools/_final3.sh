OUT=gpurun_out/r5h; mkdir -p $OUT
python bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
( for S in 0 1 2 3 4 8; do PF_OP_REPEAT=9 python tools/bench_dec_ffn.py 5344 $S; done; PF_OP_REPEAT=9 python tools/bench_dec_ffn.py 21376 0; PF_OP_REPEAT=9 python tools/bench_dec_ffn.py 1300 0 ) 2>&1 | grep splits > $OUT/dec_ffn.txt
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-via-recognizer --model seaco --in-flight 2 > $OUT/bench_seaco_2.json 2>> $OUT/bench.err
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-via-recognizer --model seaco --accuracy exact > $OUT/bench_seaco_exact.json 2>> $OUT/bench.err
python -m pytest tests/test_gpu_fp32_mode.py -x -q -m gpu -k "threshold or linear32 or ffn32" 2>&1 | tail -5 > $OUT/t.txt
cat $OUT/t.txt $OUT/dec_ffn.txt
python -c "
import json
for f in ('bench','bench_seaco_2','bench_seaco_exact'):
    try:
        d=json.load(open('$OUT/'+f+'.json')); print(f, d['ms_per_step'], d.get('ms_per_step_one_in_flight'), d['roofline'].get('traffic') if d.get('roofline') else None, d['ids_vs_fp32_oracle'])
    except Exception as e: print(f,'FAILED',e)
"
tail -3 $OUT/bench.err
