#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r6s2; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_recognizer.py tests/test_gpu_export_walk.py -x -q -m gpu -s > $O/t1.log 2>&1; echo "t1 rc=$?" >> $O/t1.log
timeout 900 python -m pytest tests/test_gpu_int8.py -x -q -m gpu -s -k "full_depth" > $O/t2.log 2>&1; echo "t2 rc=$?" >> $O/t2.log
bash tools/ffn_timing.sh > $O/timing_base.txt 2>&1
FF_EXTRA="-DFF_QK_PLAIN" bash tools/ffn_timing.sh > $O/timing_qkplain.txt 2>&1
FF_EXTRA="-DFF_XOUT_LATE" bash tools/ffn_timing.sh > $O/timing_xlate.txt 2>&1
FF_EXTRA="-DFF_XOUT_LATE -DFF_QK_PLAIN" bash tools/ffn_timing.sh > $O/timing_both.txt 2>&1
timeout 600 python bench.py --accuracy exact --steps 10 --warmup 2 --no-cpu-baseline --no-via-recognizer > $O/bench_exact.json 2> $O/bench_exact.err
tail -n 4 $O/t1.log; tail -n 6 $O/t2.log
cat $O/timing_base.txt
for f in qkplain xlate both; do echo "== $f"; grep -E "total|^ +(5|6|7|8|9|1[0-4]) " $O/timing_$f.txt; done
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6s2/bench_exact.json').read().strip().splitlines()[-1])
print('exact main path: ms/step', d['ms_per_step'], 'one in flight', d['ms_per_step_one_in_flight'], d['identical_to_fp32_oracle'])
PY
