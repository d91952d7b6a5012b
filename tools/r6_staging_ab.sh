#!/bin/bash
# same-box A/B of the recognizer's pinned staging ring (round 6): PF_RECOGNIZER_STAGING_MB=0 (the runtime's pageable copy) against
# the ring at 512 KB / 1 MB / 2 MB pieces; host float32 audio in -> texts out, 32 x 30 s per GetResults, 4 callers and one caller
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
run() {
  python bench.py --via recognizer --callers $CALLERS --steps 32 --no-cpu-baseline --no-exact $FRESH 2>gpurun_out/staging_err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])['via_recognizer']
print('   $FRESH $CALLERS callers: %.3f ms per batch   one caller: %.3f ms per batch %s  oracle: %s' % (d['ms_per_batch'], d['ms_per_batch_one_caller'], d['one_caller_split'], d['ids_vs_fp32_oracle'] and d['ids_vs_fp32_oracle']['ok']))"
}
export PF_UPLOAD_TIMING=1
for rep in 1 2; do for FRESH in "" "--fresh-host-audio"; do
  export FRESH CALLERS=4
  echo "[staging off]"; PF_RECOGNIZER_STAGING_MB=0 run; grep "^.upload" gpurun_out/staging_err.txt | tail -1
  echo "[staging always]"; PF_RECOGNIZER_STAGING_POLICY=always run; grep "^.upload" gpurun_out/staging_err.txt | tail -1
  echo "[staging auto (default)]"; run; grep "^.upload" gpurun_out/staging_err.txt | tail -1
done; done
