# A/B runs of bench.py under environment switches: bash tools/ab_inflight.sh  (results in gpurun_out/ab_*.json)
export TMPDIR=/tmp
run() { name=$1; shift; env "$@" timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline $EXTRA > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err; }
for rep in 1 2; do
EXTRA="--model sensevoice" run s_base_$rep A=1
EXTRA="--model sensevoice" run s_bigp2_$rep PF_BIGP=2
EXTRA="--model sensevoice" run s_old_$rep PF_QKV_FILL=85
EXTRA="--model sensevoice --in-flight 1" run s1_base_$rep A=1
EXTRA="--batch 128 --steps 6" run b128_base_$rep A=1
EXTRA="--accuracy int8" run i8_base_$rep A=1
done
