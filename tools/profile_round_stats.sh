#!/bin/bash
# The rocprofv3 part of tools/profile_round.sh alone (kernel stats with one / two steps in flight, roofline table, FETCH / WRITE PMC passes)
# against an existing gpurun_out/$TAG/bench.json — for re-collecting the traces after a kernel change without re-running every bench line.
TAG=${1:-round6}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
[ -f $OUT/bench.json ] || python bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats2 -- python $OLDPWD/bench.py --no-cpu-baseline --no-via-recognizer --no-exact --steps 6 --warmup 2 ) > $OUT/rocprof_stats2.log 2>&1
find /tmp/prof_stats2 -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_two_in_flight.csv \;
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python $OLDPWD/bench.py --no-cpu-baseline --no-via-recognizer --no-exact --steps 5 --warmup 2 --in-flight 1 ) > $OUT/rocprof_stats.log 2>&1
find /tmp/prof_stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find /tmp/prof_stats -name "*kernel_trace.csv" -exec cp {} /tmp/kernel_trace.csv \;
python tools/roofline_table.py /tmp/kernel_trace.csv $OUT/bench.json $TAG > $OUT/roofline.md 2>> $OUT/bench.err
python tools/step_gaps.py /tmp/kernel_trace.csv > $OUT/step_gaps.txt 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/prof_$C
  ( cd /tmp && rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/prof_$C -- python $OLDPWD/bench.py --no-cpu-baseline --no-via-recognizer --no-exact --steps 1 --warmup 1 --in-flight 1 ) > $OUT/rocprof_$C.log 2>&1
  find /tmp/prof_$C -name "*counter_collection.csv" -exec cp {} /tmp/pmc_$C.csv \;
done
python tools/pmc_summary.py /tmp/pmc_FETCH_SIZE.csv /tmp/pmc_WRITE_SIZE.csv > $OUT/pmc.json 2>> $OUT/bench.err
head -14 $OUT/kernel_stats.csv | cut -c1-160
