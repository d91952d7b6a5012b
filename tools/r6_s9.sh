#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
python -m pytest tests/test_gpu_pipeline_kernels.py tests/test_gpu_interference.py -x -q -m gpu 2>&1 | tail -5
