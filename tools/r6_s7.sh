#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r6s7; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_pipeline_kernels.py -x -q -m gpu -s -k "decoder_middle" > $O/t1.log 2>&1; echo "t1 rc=$?" >> $O/t1.log
tail -n 12 $O/t1.log
for rep in 1 2; do for f in 0 1; do
  for e in 1 2; do
  PF_DEC_MID=$f python bench.py --steps 30 --warmup 5 --in-flight $e --no-cpu-baseline --no-via-recognizer --no-exact 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
c=d['class_ms_per_step']
print('PF_DEC_MID=$f in-flight $e: ms/step %.3f' % d['ms_per_step'], {k:round(c[k]['ms'],3) for k in ('gemm_dec_ffn','fsmn','gemm_dec_q','dec_mid','attn_cross','layernorm') if k in c}, d['ids_vs_fp32_oracle']['agree_all_positions'])"
  done
done; done
