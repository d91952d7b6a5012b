#!/bin/bash
# same-box A/B: copy lanes at default vs highest stream priority, 4 callers, fresh host arrays (every upload staged) and same arrays
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp PF_UPLOAD_TIMING=1
run() {
  python bench.py --via recognizer --callers 4 --steps 48 --no-cpu-baseline --no-exact $FRESH 2>gpurun_out/lane_err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])['via_recognizer']
print('   $FRESH 4 callers: %.3f ms per batch   one caller: %.3f' % (d['ms_per_batch'], d['ms_per_batch_one_caller']))"
  grep "^.upload" gpurun_out/lane_err.txt | tail -1
}
for rep in 1 2 3; do for FRESH in "--fresh-host-audio" ""; do
  export FRESH
  echo "[lane priority default]"; PF_RECOGNIZER_LANE_PRIORITY=0 run
  echo "[lane priority high]"; PF_RECOGNIZER_LANE_PRIORITY=1 run
done; done
