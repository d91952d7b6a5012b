export TMPDIR=/tmp
OUT=gpurun_out/r5g; mkdir -p $OUT
python bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
bash tools/feed_gap.sh r5g_fg > $OUT/feed_gap.log 2>&1
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p2 -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-via-recognizer --steps 4 --warmup 1 --accuracy exact ) > $OUT/p2.log 2>&1
find /tmp/p2 -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_exact.csv \;
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p3 -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-via-recognizer --steps 4 --warmup 1 --accuracy int8 --in-flight 1 ) > $OUT/p3.log 2>&1
find /tmp/p3 -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_int8.csv \;
python tools/bench_dec_ffn.py 5344 0 > $OUT/dec_ffn.txt 2>&1
for S in 1 2 3 4 8; do python tools/bench_dec_ffn.py 5344 $S >> $OUT/dec_ffn.txt 2>&1; done
python tools/bench_dec_ffn.py 21376 0 >> $OUT/dec_ffn.txt 2>&1
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-via-recognizer --model sensevoice --accuracy exact > $OUT/bench_sensevoice_exact.json 2>> $OUT/bench.err
ls $OUT gpurun_out/r5g_fg; python -c "
import json; d=json.load(open('$OUT/bench.json')); print(d['ms_per_step'], d['roofline']['traffic'], d['cpu_baseline']['value'])"
