#!/bin/bash
# same-box A/B of compile-time variants of ONE csrc file: tools/r6_ab2.sh <file> "<flags A>" "<flags B>" ...
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
F=$1; shift
export TMPDIR=/tmp
run() {
  for e in 1 2; do
    python bench.py --steps 30 --warmup 5 --in-flight $e --no-cpu-baseline --no-via-recognizer --no-exact 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('   in-flight $e: ms/step %.3f  dominant %.1f us' % (d['ms_per_step'], d['roofline']['avg_us']), {k:round(v['ms'],3) for k,v in d['class_ms_per_step'].items() if v['ms']>0.3}, d['ids_vs_fp32_oracle']['agree_all_positions'])"
  done
}
for rep in 1 2; do
  for V in "$@"; do
    touch aliparaformerasr_amd/csrc/$F; make -C aliparaformerasr_amd/csrc EXTRA="$V" > /dev/null 2>&1 || echo BUILD-FAILED
    echo "[$V] ($rep)"; run
  done
done
touch aliparaformerasr_amd/csrc/$F; make -C aliparaformerasr_amd/csrc > /dev/null 2>&1
