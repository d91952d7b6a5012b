#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r6s4; mkdir -p $O
export TMPDIR=/tmp
timeout 300 tools/ubench/ldbw4 > $O/ldbw4.txt 2>&1
FF_EXTRA="-DFF_XOUT_LATE -DFF_QK_PLAIN -DFF_PASS_ROT" bash tools/ffn_timing.sh > $O/timing_rot.txt 2>&1
cat $O/ldbw4.txt
grep -E "total|^ +(7|8|9|1[0-4]) " $O/timing_rot.txt
