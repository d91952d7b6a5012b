#!/bin/bash
# A/B of the persistent BiLSTM's h-load form in ONE session (0 = one wait per load [default], 1 = eight loads, one wait)
# and against the per-step launches (PF_LSTM_STEPS=1); prints ms/step of configs[4] and the recurrence's share
for i in 1 2; do for v in 0 1; do PF_LSTM_VAR=$v timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --model seaco --breakdown 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('var $v', d['ms_per_step'], d['kernel_breakdown_ms_per_step']['lstm']['ms'])"; done; done
PF_LSTM_STEPS=1 timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --model seaco --breakdown 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('per-step launches', d['ms_per_step'], d['kernel_breakdown_ms_per_step']['lstm']['ms'])"
