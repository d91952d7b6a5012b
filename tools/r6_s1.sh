#!/bin/bash
# round 6, GPU session 1: new tests, default bench line (with the exact object), half-batch concurrency experiment
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r6s1; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_recognizer.py tests/test_gpu_harness.py -x -q -m gpu > $O/t1.log 2>&1; echo "t1 rc=$?" >> $O/t1.log
timeout 1500 python -m pytest tests/test_gpu_full_depth.py -x -q -m gpu -k "exact" -s > $O/t2.log 2>&1; echo "t2 rc=$?" >> $O/t2.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "rc=$?" >> $O/bench_default.err
for cfg in "16 2" "16 4" "16 1" "32 1" "8 4"; do
  set -- $cfg
  timeout 300 python bench.py --batch $1 --in-flight $2 --steps 20 --warmup 4 --no-cpu-baseline --no-via-recognizer --no-exact > $O/bench_b$1_e$2.json 2> $O/bench_b$1_e$2.err
done
tail -3 $O/t1.log $O/t2.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6s1/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        ex=d.get('exact') or {}
        print(f.split('/')[-1], 'ms/step', round(d['ms_per_step'],3), 'serial', round(d.get('ms_per_step_one_in_flight') or 0,3), 'roof', round(d['roofline']['frac'],4),
              'exact', ex.get('ms_per_step'), ex.get('ms_per_step_one_in_flight'), ex.get('identical_to_fp32_oracle'), (ex.get('roofline') or {}).get('frac'),
              'via1', (d.get('via_recognizer') or {}).get('ms_per_batch_one_caller'))
    except Exception as e:
        print(f, 'ERR', e)
PY
