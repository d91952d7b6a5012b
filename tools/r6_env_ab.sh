#!/bin/bash
# same-box A/B of one environment knob through the engine bench (one and two steps in flight) and the recognizer line:
#   tools/r6_env_ab.sh NAME VALUE_A VALUE_B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
K=$1; A=$2; B=$3
run() {
  for e in 1 2; do
    python bench.py --steps 30 --warmup 6 --in-flight $e --no-cpu-baseline --no-via-recognizer --no-exact 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('   in-flight $e: ms/step %.3f' % d['ms_per_step'])"
  done
  python bench.py --via recognizer --callers 4 --steps 48 --no-cpu-baseline --no-exact 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])['via_recognizer']
print('   recognizer, 4 callers: %.3f ms per batch   one caller: %.3f' % (d['ms_per_batch'], d['ms_per_batch_one_caller']))"
}
for rep in 1 2 3; do
  echo "[$K=$A] ($rep)"; env $K=$A bash -c "$(declare -f run); run"
  echo "[$K=$B] ($rep)"; env $K=$B bash -c "$(declare -f run); run"
done
