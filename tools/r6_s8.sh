#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
touch aliparaformerasr_amd/csrc/k_decmid.hip
make -C aliparaformerasr_amd/csrc EXTRA="-DDM_TIMING" > /dev/null 2>&1 || echo BUILD-FAILED
python tools/decmid_timing.py
touch aliparaformerasr_amd/csrc/k_decmid.hip; make -C aliparaformerasr_amd/csrc > /dev/null 2>&1
