#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r6s5; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_pipeline_kernels.py tests/test_gpu_interference.py tests/test_gpu_recognizer.py tests/test_gpu_export_walk.py tests/test_gpu_harness.py -x -q -m gpu > $O/t1.log 2>&1; echo "t1 rc=$?" >> $O/t1.log
timeout 1500 python -m pytest tests/test_gpu_full_depth.py -x -q -m gpu > $O/t2.log 2>&1; echo "t2 rc=$?" >> $O/t2.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "rc=$?" >> $O/bench_default.err
bash tools/ffn_timing.sh > $O/timing_head.txt 2>&1
tail -n 5 $O/t1.log; tail -n 5 $O/t2.log; tail -n 3 $O/bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6s5/bench_default.json').read().strip().splitlines()[-1])
ex=d.get('exact') or {}
print('ms/step', round(d['ms_per_step'],3), 'serial', round(d.get('ms_per_step_one_in_flight') or 0,3), 'roof', d['roofline']['frac'], d['roofline']['avg_us'],
      'exact', ex.get('ms_per_step'), ex.get('ms_per_step_one_in_flight'), ex.get('identical_to_fp32_oracle'),
      'via', (d.get('via_recognizer') or {}).get('ms_per_batch'), (d.get('via_recognizer') or {}).get('ms_per_batch_one_caller'), d['ids_vs_fp32_oracle'])
print({k:v['ms'] for k,v in d['class_ms_per_step'].items()})
PY
cat $O/timing_head.txt
