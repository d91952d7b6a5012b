#!/bin/bash
# same-box A/B of PF_RECOGNIZER_STAGGER (fraction of the recent step duration; 0 = off) on the recognizer lines: 2 and 4 callers, same / fresh host arrays
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
run() {
  python bench.py --via recognizer --callers $1 --steps 48 --no-cpu-baseline --no-exact $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])['via_recognizer']
print('   $1 callers $2: %.3f ms per batch   one caller: %.3f' % (d['ms_per_batch'], d['ms_per_batch_one_caller']))"
}
for rep in 1 2; do for us in 0 0.25 0.35 0.45; do
  echo "[PF_RECOGNIZER_STAGGER=$us] ($rep)"
  export PF_RECOGNIZER_STAGGER=$us
  run 2 ""; run 4 ""; run 3 ""; run 2 "--fresh-host-audio"; run 4 "--fresh-host-audio"
done; done
