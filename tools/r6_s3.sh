#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r6s3; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_recognizer.py tests/test_gpu_export_walk.py -x -q -m gpu -s > $O/t1.log 2>&1; echo "t1 rc=$?" >> $O/t1.log
FF_EXTRA="-DFF_XOUT_LATE -DFF_QK_PLAIN -DFF_PFO=16 -DFF_PFQ=12" bash tools/ffn_timing.sh > $O/timing_v1.txt 2>&1
FF_EXTRA="-DFF_QK_PLAIN -DFF_PFO=16 -DFF_PFQ=16" bash tools/ffn_timing.sh > $O/timing_v2.txt 2>&1
FF_EXTRA="-DFF_XOUT_LATE -DFF_QK_PLAIN -DFF_PFO=16" bash tools/ffn_timing.sh > $O/timing_v3.txt 2>&1
tail -n 8 $O/t1.log
for f in v1 v2 v3; do echo "== $f"; cat $O/timing_$f.txt; done
