#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
python -m pytest tests/test_gpu_recognizer.py tests/test_gpu_harness.py tests/test_gpu_seaco.py tests/test_gpu_sensevoice.py -x -q -m gpu 2>&1 | tail -4
for rep in 1 2; do for n in 0 2 4 8; do
  PF_RECOGNIZER_UPLOAD_THREADS=$n python bench.py --via recognizer --steps 24 --callers 1 --in-flight 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); v=d['via_recognizer']
print('upload threads $n: one caller %.3f ms per batch (timed loop %.3f)' % (v['ms_per_batch_one_caller'], v['ms_per_batch']))"
done; done
for n in 0 4; do
  PF_RECOGNIZER_UPLOAD_THREADS=$n python bench.py --via recognizer --steps 32 --callers 4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); v=d['via_recognizer']
print('upload threads $n, 4 callers: %.3f ms per batch' % v['ms_per_batch'])"
done
