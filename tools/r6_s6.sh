#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r6s6; mkdir -p $O
export TMPDIR=/tmp
FF_EXTRA="-DFF_P1_ROT" bash tools/ffn_timing.sh > $O/timing_p1rot.txt 2>&1
FF_EXTRA="-DFF_TIMING_MAIN" FF_PY=tools/ffn_timing_main.py bash tools/ffn_timing.sh > $O/timing_main.txt 2>&1
grep -E "total|^ +(0|1|2|3|4) " $O/timing_p1rot.txt
cat $O/timing_main.txt
