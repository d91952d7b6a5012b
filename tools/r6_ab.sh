#!/bin/bash
# same-box A/B of two versions of one csrc file: tools/r6_ab.sh <file under csrc> <old version path> [bench args]
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
F=$1; OLD=$2; shift 2
O=gpurun_out/r6ab; mkdir -p $O
export TMPDIR=/tmp
cp aliparaformerasr_amd/csrc/$F /tmp/new_$F
run() {
  for e in 1 2; do
    python bench.py --steps 30 --warmup 5 --in-flight $e --no-cpu-baseline --no-via-recognizer --no-exact "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('   in-flight $e: ms/step %.3f  dominant %.1f us  classes' % (d['ms_per_step'], d['roofline']['avg_us']), {k:round(v['ms'],3) for k,v in d['class_ms_per_step'].items() if v['ms']>0.2})"
  done
}
for rep in 1 2; do
  cp $OLD aliparaformerasr_amd/csrc/$F; make -C aliparaformerasr_amd/csrc > /dev/null 2>&1 || echo BUILD-FAILED
  echo "OLD ($rep)"; run "$@"
  cp /tmp/new_$F aliparaformerasr_amd/csrc/$F; touch aliparaformerasr_amd/csrc/$F; make -C aliparaformerasr_amd/csrc > /dev/null 2>&1 || echo BUILD-FAILED
  echo "NEW ($rep)"; run "$@"
done
