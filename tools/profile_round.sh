#!/bin/bash
# Round profile on the GPU box (run through gpurun): bench lines of every config, rocprofv3 kernel stats of the headline
# command, PMC traffic (separate passes, per the MI355X guide) of the dominant kernels.  Everything lands in gpurun_out/$TAG.
TAG=${1:-round6}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-via-recognizer --in-flight 1 > $OUT/bench_one_in_flight.json 2>> $OUT/bench.err
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-via-recognizer --accuracy int8 > $OUT/bench_int8.json 2>> $OUT/bench.err
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-via-recognizer --model seaco --accuracy int8 > $OUT/bench_seaco_int8.json 2>> $OUT/bench.err
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-via-recognizer --accuracy fp32 > $OUT/bench_fp32.json 2>> $OUT/bench.err
python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-via-recognizer --accuracy exact > $OUT/bench_exact.json 2>> $OUT/bench.err
python bench.py --via recognizer --steps 32 --callers 4 > $OUT/bench_via_recognizer_4callers.json 2>> $OUT/bench.err
python bench.py --via recognizer --steps 24 --callers 2 --in-flight 2 > $OUT/bench_via_recognizer_2callers.json 2>> $OUT/bench.err
python bench.py --via recognizer --steps 32 --callers 4 --fresh-host-audio > $OUT/bench_via_recognizer_4callers_fresh.json 2>> $OUT/bench.err
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-via-recognizer --model sensevoice > $OUT/bench_sensevoice.json 2>> $OUT/bench.err
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-via-recognizer --model seaco > $OUT/bench_seaco.json 2>> $OUT/bench.err
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-via-recognizer --batch 128 > $OUT/bench_batch128.json 2>> $OUT/bench.err
python bench.py --group 4 --group-devices 0,0,0,0 --batch 32 --steps 5 --warmup 2 > $OUT/bench_group4x32.json 2>> $OUT/bench.err
python tools/latency.py > $OUT/latency.txt 2>> $OUT/bench.err
python tools/bench_online.py > $OUT/online.txt 2>> $OUT/bench.err
# kernel trace and counters with ONE step in flight: the durations are then the kernels' own (bench.py event-times its dominant
# class in a step that runs alone, too); a second trace with the default two steps in flight shows what co-scheduling does to them
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats2 -- python $OLDPWD/bench.py --no-cpu-baseline --no-via-recognizer --no-exact --steps 6 --warmup 2 ) > $OUT/rocprof_stats2.log 2>&1
find /tmp/prof_stats2 -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_two_in_flight.csv \;
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python $OLDPWD/bench.py --no-cpu-baseline --no-via-recognizer --no-exact --steps 5 --warmup 2 --in-flight 1 ) > $OUT/rocprof_stats.log 2>&1
( cd $GRAFT_REPO_ROOT 2>/dev/null || true )
find /tmp/prof_stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find /tmp/prof_stats -name "*kernel_trace.csv" -exec cp {} /tmp/kernel_trace.csv \;
python tools/roofline_table.py /tmp/kernel_trace.csv $OUT/bench.json $TAG > $OUT/roofline.md 2>> $OUT/bench.err
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/prof_$C
  ( cd /tmp && rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/prof_$C -- python $OLDPWD/bench.py --no-cpu-baseline --no-via-recognizer --no-exact --steps 1 --warmup 1 --in-flight 1 ) > $OUT/rocprof_$C.log 2>&1
  find /tmp/prof_$C -name "*counter_collection.csv" -exec cp {} /tmp/pmc_$C.csv \;
done
python tools/pmc_summary.py /tmp/pmc_FETCH_SIZE.csv /tmp/pmc_WRITE_SIZE.csv > $OUT/pmc.json 2>> $OUT/bench.err
ls -la $OUT
tail -3 $OUT/bench.err
python -c "
import json
for f in ('bench','bench_one_in_flight','bench_sensevoice','bench_seaco','bench_batch128','bench_int8','bench_seaco_int8','bench_fp32','bench_exact','bench_via_recognizer_4callers','bench_via_recognizer_2callers','bench_via_recognizer_4callers_fresh'):
    try:
        d=json.load(open('$OUT/'+f+'.json')); print(f, round(d['ms_per_step'],3), round(d['value']), d['roofline']['frac'], d.get('cpu_baseline',{}).get('value'))
    except Exception as e: print(f, 'FAILED', e)
"
cat $OUT/latency.txt $OUT/online.txt
