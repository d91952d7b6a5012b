export TMPDIR=/tmp
ROOT=$PWD
for name in a b; do
  if [ $name = a ]; then L="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"; else L="SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM"; fi
  rm -rf /tmp/prof_$name
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $L --output-format csv -d /tmp/prof_$name -- python $ROOT/bench.py --no-cpu-baseline --no-via-recognizer --no-exact --steps 2 --warmup 1 --in-flight 1 ) > gpurun_out/fbank_pmc_$name.log 2>&1
  find /tmp/prof_$name -name "*counter_collection.csv" -exec cp {} /tmp/pmc_$name.csv \;
  python tools/pmc_counters.py /tmp/pmc_$name.csv 1 | python -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d.items():
    if 'fbank' in k: print(k[:30], json.dumps(v))"
done
