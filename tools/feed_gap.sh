#!/bin/bash
# Counters behind the "operand feed gap" question: SQ wait / issue / LDS / MFMA-busy cycles and TA / TCP
# stall counters of the encoder GEMM kernels, one rocprofv3 --pmc pass per counter group (never combined with tracing
# domains).  Output: gpurun_out/$TAG/pmc_<group>.json (tools/pmc_counters.py)
TAG=${1:-round6}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
( cd /tmp && rocprofv3 -L > $ROOT/$OUT/counters_available.txt 2>&1 )
have() { grep -qw "$1" $OUT/counters_available.txt; }
run_pass() {
  local name=$1; shift
  local list=""
  for c in "$@"; do if have $c; then list="$list $c"; else echo "counter $c not available" >> $OUT/pmc_skipped.txt; fi; done
  [ -z "$list" ] && return
  rm -rf /tmp/prof_$name
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $list --output-format csv -d /tmp/prof_$name -- python $ROOT/bench.py --no-cpu-baseline --no-via-recognizer --no-exact --steps 1 --warmup 1 --in-flight 1 ) > $OUT/rocprof_$name.log 2>&1
  find /tmp/prof_$name -name "*counter_collection.csv" -exec cp {} /tmp/pmc_$name.csv \;
  python tools/pmc_counters.py /tmp/pmc_$name.csv > $OUT/pmc_$name.json 2>> $OUT/pmc_err.txt
}
run_pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES
run_pass sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_LDS
run_pass ta TA_TA_BUSY_sum TA_BUFFER_LOAD_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
run_pass tcp TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum
run_pass tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum
run_pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
ls -la $OUT
